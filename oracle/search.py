"""Query-scoring oracle -- test infrastructure, see oracle/__init__.py.

Restates webui.py:63-80 (filter_searched_result, PINNED by tests/golden/g5_filter.json),
webui.py:82-117 (query vector), webui.py:345-383 (parse + combine) and
webui.py:189-253 (rerank) in numpy.  The dense index product
(gensim Similarity.__getitem__, absent here -> PARITY UNPINNED) is defined as the
k-ordered float32 fmaf chain of oracle/csrc/oracle.c::orc_sim_chain.
"""
import ctypes
import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import bm25 as obm25

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

BM25_WEIGHT = 0.5               # webui.py:51
DOC2VEC_WEIGHT = 0.5            # webui.py:52
ORIGINAL_SCORE_WEIGHT = 0.7     # webui.py:55
RERANKED_SCORE_WEIGHT = 0.3     # webui.py:56
DIFF_FILTER_THRESH = 1e-6       # webui.py:58
REQUIRE_TAG_MAGIC_NUMBER = 1000  # webui.py:60


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = ctypes.CDLL(path)
    return _lib


def similarity(index: np.ndarray, q: np.ndarray) -> np.ndarray:
    """scores[d] = fmaf-chain_k index[d,k]*q[k]  (float32)."""
    index = np.ascontiguousarray(index, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    D, K = index.shape
    out = np.empty(D, dtype=np.float32)
    lib().orc_sim_chain(index.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(D), ctypes.c_int(K),
                        ctypes.c_int64(K), q.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def filter_searched_result(sorted_scores: List[Tuple[int, float]]) -> List[Tuple[int, float]]:
    """webui.py:63-80."""
    scores = np.array([s for _, s in sorted_scores])
    diff = scores[:-1] - scores[1:]
    diff = np.where(diff == 0, np.inf, diff)                 # :70
    t = len(sorted_scores)                                   # :72
    found = np.where(diff < DIFF_FILTER_THRESH)[0]           # :73
    if len(found) == 1:
        t = found[0]
    elif len(found) >= 2:
        t = found[1]                                         # :77 second point
    max_val = scores.max()
    return [(sorted_scores[i][0], sorted_scores[i][1] / float(max_val))
            for i in range(int(t)) if sorted_scores[i][1] > 0]   # :80


def parse_query(new_doc: str):
    """Both parsers of the reference, kept separate because they differ:
    returns (d2v_terms [(tag_with_escaped_parens, int weight)], all_weight,
             bm25_terms [(tag, kind, int weight)] with kind in {'plain','require','exclude'})."""
    d2v_terms: List[Tuple[str, int]] = []
    all_weight = 0
    for tag in new_doc.split(" "):                           # webui.py:83-99
        sp = tag.split(":")
        if len(sp) >= 2 and (sp[-1].startswith("+") or sp[-1].startswith("-") or sp[-1].isdigit()):
            elem = ":".join(sp[:-1]).replace("\\(", "(").replace("\\)", ")")
            d2v_terms.append((elem.replace("(", "\\(").replace(")", "\\)"), int(sp[-1])))
            all_weight += int(sp[-1])
        else:
            elem = ":".join(sp).replace("\\(", "(").replace("\\)", ")")
            d2v_terms.append((elem.replace("(", "\\(").replace(")", "\\)"), 1))
            all_weight += 1
    if all_weight == 0:
        all_weight = 1                                       # :101-102
    bm25_terms: List[Tuple[str, str, int]] = []
    for term in new_doc.split(" "):                          # webui.py:354-371
        sp = term.split(":")
        if len(sp) >= 2 and (sp[-1].startswith("+") or sp[-1].startswith("-") or sp[-1].isdigit()):
            if sp[-1].startswith("+"):
                bm25_terms.append((":".join(sp[:-1]), "require", int(sp[-1])))
            else:
                bm25_terms.append((":".join(sp[:-1]), "exclude", int(sp[-1])))
        else:
            bm25_terms.append((":".join(sp), "plain", 1))
    return d2v_terms, all_weight, bm25_terms


def query_vector(d2v_terms, all_weight, infer: Callable[[List[str]], np.ndarray], dim: int) -> np.ndarray:
    """webui.py:104-117 -> float64[dim] unit vector."""
    got = np.zeros(dim)
    for tag, weight in d2v_terms:
        v = infer([tag])
        v = v / np.linalg.norm(v)
        got += weight * v
    got = got / all_weight
    norm = np.linalg.norm(got)
    if math.isinf(norm) or norm == 0:
        norm = 1.0
    return got / norm


def query_weights(bm25_terms, token2id: Dict[str, int]) -> Dict[int, float]:
    """webui.py:355-371 (KeyError for unknown tags, like the reference)."""
    qw: Dict[int, float] = {}
    for tag, kind, w in bm25_terms:
        if kind == "require":
            qw[token2id[tag]] = REQUIRE_TAG_MAGIC_NUMBER + w
        elif kind == "exclude":
            qw[token2id[tag]] = w
        else:
            qw[token2id[tag]] = 1
    return qw


def combine(bm25_scores: np.ndarray, sims: np.ndarray) -> np.ndarray:
    """webui.py:377-383."""
    if sims.max() > 0:
        sims = sims / sims.max()
    if bm25_scores.max() > 0:
        bm25_scores = bm25_scores / bm25_scores.max()
    return BM25_WEIGHT * bm25_scores + DOC2VEC_WEIGHT * sims


def stable_rank(final_scores: np.ndarray) -> np.ndarray:
    """ids in the order of sorted(enumerate(scores), key=-score)  (webui.py:191-192):
    score descending, ties by ascending doc id (Python's sort is stable)."""
    return np.lexsort((np.arange(len(final_scores)), -final_scores))


def topk(final_scores: np.ndarray, k: int):
    order = stable_rank(final_scores)[:k]
    return order, final_scores[order]


def rerank(final_scores: np.ndarray, topn: int, rerank_sims_fn: Callable[[np.ndarray, np.ndarray], np.ndarray],
           ) -> List[Tuple[int, float]]:
    """webui.py:189-253 'normal' mode.  rerank_sims_fn(top10_ids, top10_scores) must return
    the float32[D] similarity of the index with the pseudo-relevance query built from the
    top-10 documents (webui.py:198-205; see hiptagsearch.search for both the intended and
    the bug-compatible construction of that query)."""
    order = stable_rank(final_scores)
    sims = [(int(i), float(final_scores[i])) for i in order]
    if len(sims) > 10:
        top10 = sims[:10]
        top10_ids = [d for d, _ in top10]
        top10_set = set(top10_ids)
        rs = rerank_sims_fn(np.array(top10_ids), np.array([s for _, s in top10]))
        rf = ORIGINAL_SCORE_WEIGHT * final_scores + RERANKED_SCORE_WEIGHT * rs       # :208
        if rf.max() > 0:
            rf = rf / rf.max()                                                        # :210-211
        rest_order = stable_rank(rf)
        final = [(d, 1.0) for d in top10_ids]                                         # :219-222
        final += [(int(i), float(rf[i])) for i in rest_order if int(i) not in top10_set]   # :217,225,228-237
        final = filter_searched_result(final)                                         # :240
        return final[:min(topn, len(final))]
    sims = filter_searched_result(sims)
    return sims[:min(topn, len(sims))]


def cfeatures_rerank(final_scores: np.ndarray, topn: int, required_tags: Sequence[str], exclude_tags: Sequence[str],
                     image_files_name_tags_arr: Sequence[str], cfeature_filepath_idx: Sequence[str], cfeature_rows: np.ndarray,
                     get_image_feature: Callable[[str], Optional[np.ndarray]], threshold: float) -> List[Tuple[int, float]]:
    """webui.py:255-342 'character oriented' mode, with the CCIP metric model's difference (an opaque ONNX graph,
    gen_cfeatures.py:212-274) restated as BASELINE.json configs[4] does: difference := 1 - cosine(feature row, query), the
    rows being the unit vectors the feature index stores (gen_cfeatures.py:310-314) and the product the k-ordered chain of
    similarity().  `get_image_feature(path)` stands for predictor.get_image_feature (:296-301; None = failed load)."""
    file_tag_index_dict = {l.split(",")[0]: {t: True for t in l.split(",")[1:]} for l in image_files_name_tags_arr}   # webui.py:623-646
    filepath_docid_dict = {l.split(",")[0]: i for i, l in enumerate(image_files_name_tags_arr)}
    order = stable_rank(final_scores)                                                   # :282-283
    sims = [(int(i), float(final_scores[i])) for i in order]
    if len(sims) > 10:
        top10 = sims[:10]
        feats = []
        for doc_id, _ in top10:                                                         # :289-301
            f = get_image_feature(image_files_name_tags_arr[doc_id].split(",")[0])
            if f is not None:
                feats.append(f)
        mean = np.average(np.stack(feats), axis=0).astype(np.float32)                   # :303
        n = np.float32(np.sqrt(np.sum(mean * mean)))
        q = mean / n if n > 0 else mean
        diffs = np.float32(1.0) - similarity(cfeature_rows, q)                          # :306-309 restated as cosine
        out = []
        for idx, path in enumerate(cfeature_filepath_idx):                              # :311-328
            if path not in file_tag_index_dict:
                continue
            tags = file_tag_index_dict[path]
            if diffs[idx] < threshold and all(t in tags for t in required_tags) and all(t not in tags for t in exclude_tags):
                out.append((filepath_docid_dict[path], float(np.float32(1.0) - diffs[idx])))
        return top10 + sorted(out, key=lambda it: -it[1])                               # :330-335
    sims = filter_searched_result(sims)                                                 # :336-342
    return sims[:min(topn, len(sims))]
