"""Query function: host-side mirror of webui.py:63-390 over the device kernels.

  filter_searched_result                webui.py:63-80
  normalize_and_apply_weight_doc2vec    webui.py:82-117
  find_similar_documents                webui.py:345-390   (same name / arguments / return value)
  get_doc2vec_based_reranked_scores     webui.py:189-253

The reference keeps model, index, dictionary and the BM25 statistics in module globals that
`load_model()` fills (webui.py:649-689); here they live in a SearchEngine, and the module-level
functions of the same names operate on the engine installed with `set_engine()`.
Scoring (BM25, index product, normalise, combine, top-k) runs in libhip_tagsearch.so; parsing,
the 10-document pseudo-relevance bookkeeping and the gap filter are host logic.
"""
import ctypes
import itertools
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import c_double, c_int64
from .bm25 import BM25Index
from .d2v import Doc2VecInference
from .index import Similarity

BM25_WEIGHT: float = 0.5               # webui.py:51
DOC2VEC_WEIGHT: float = 0.5            # webui.py:52
ORIGINAL_SCORE_WEIGHT: float = 0.7     # webui.py:55
RERANKED_SCORE_WEIGHT: float = 0.3     # webui.py:56
DIFF_FILTER_THRESH = 1e-6              # webui.py:58
REQUIRE_TAG_MAGIC_NUMBER = 1000        # webui.py:60
TOPK_MAX = 1024                        # hipts_topk limit


def filter_searched_result(ranked: List[Tuple[int, float]]) -> List[Tuple[int, float]]:
    """The result filter of webui.py:63-80 on a list ranked by descending score: the list ends at the SECOND place where two neighbours
    are closer than 1e-6 without being equal (at the first if there is only one), entries with a non-positive score are dropped, and
    the scores are divided by the largest one.  Pinned by tests/golden/g5_filter.json (vectors produced by the reference's function)."""
    vals = np.fromiter((score for _, score in ranked), dtype=np.float64, count=len(ranked))
    gaps = vals[:-1] - vals[1:]
    close = np.flatnonzero((gaps != 0) & (gaps < DIFF_FILTER_THRESH))      # equal neighbours (gap 0) do not count; NaN gaps neither
    cut = len(ranked) if close.size == 0 else int(close[min(1, close.size - 1)])
    top = float(vals.max())
    return [(doc, score / top) for doc, score in ranked[:cut] if score > 0]


def _split_weight(tag: str):
    sp = tag.split(":")
    if len(sp) >= 2 and (sp[-1].startswith("+") or sp[-1].startswith("-") or sp[-1].isdigit()):
        return ":".join(sp[:-1]), sp[-1]
    return ":".join(sp), None


class SearchEngine:
    def __init__(self, model: Doc2VecInference, index: Similarity, token2id: Dict[str, int], bm25: BM25Index,
                 image_files_name_tags_arr: Sequence[str], search_mode: str = "normal", compat_rerank: bool = False):
        self.model, self.index, self.token2id, self.bm25 = model, index, token2id, bm25
        self.image_files_name_tags_arr = list(image_files_name_tags_arr)
        self.search_mode = search_mode
        self.compat_rerank = compat_rerank
        self.cindex = None                     # cfeatures.CharacterFeatureIndex for 'character oriented' mode
        self.stats = {"queries": 0, "full_rank_fallbacks": 0, "rank_continuations": 0}     # full_rank_fallbacks: always 0 since round 3
        self._search_fn = _lib.load().hipts_search
        self._w_bm25, self._w_sim = c_double(BM25_WEIGHT), c_double(DOC2VEC_WEIGHT)
        # webui.py:623-646: path -> {tag: True} and path -> doc id, built from the same index file
        self.file_tag_index_dict = {l.split(",")[0]: {t: True for t in l.split(",")[1:]} for l in self.image_files_name_tags_arr}
        self.filepath_docid_dict = {l.split(",")[0]: i for i, l in enumerate(self.image_files_name_tags_arr)}

    # ---- webui.py:82-117 ------------------------------------------------------------------------
    def normalize_and_apply_weight_doc2vec(self, new_doc: str) -> np.ndarray:
        tag_and_weight: List[Tuple[str, int]] = []
        all_weight = 0
        for tag in new_doc.split(" "):
            name, w = _split_weight(tag)
            name = name.replace("\\(", "(").replace("\\)", ")").replace("(", "\\(").replace(")", "\\)")   # :92-98
            wi = int(w) if w is not None else 1
            tag_and_weight.append((name, wi))
            all_weight += wi
        if all_weight == 0:
            all_weight = 1
        vecs = self.model.infer_vectors([[t] for t, _ in tag_and_weight])            # :106, batched
        got = np.zeros(self.model.vector_size)
        for (tag, weight), v in zip(tag_and_weight, vecs):
            v = v / np.linalg.norm(v)                                                # :107
            got += weight * v                                                        # :108
        got = got / all_weight
        norm = np.linalg.norm(got)
        if math.isinf(norm) or norm == 0:
            norm = 1.0
        return got / norm

    # ---- webui.py:354-371 -----------------------------------------------------------------------
    def parse_bm25_query(self, new_doc: str):
        qw: Dict[int, float] = {}
        required: List[str] = []
        exclude: List[str] = []
        for term in new_doc.split(" "):
            name, w = _split_weight(term)
            if w is not None:
                if w.startswith("+"):
                    qw[self.token2id[name]] = REQUIRE_TAG_MAGIC_NUMBER + int(w)      # KeyError like :364
                    required.append(name)
                else:
                    qw[self.token2id[name]] = int(w)
                    exclude.append(name)
            else:
                qw[self.token2id[name]] = 1
        return qw, required, exclude

    # ---- device scoring -------------------------------------------------------------------------
    def score_topk(self, query_weights: Sequence[Dict[int, float]], query_vectors: np.ndarray, k: int,
                   final_out=None) -> Tuple[np.ndarray, np.ndarray]:
        """Fused webui.py:352-383 for a batch: returns (ids int32 [nq,k], scores float64 [nq,k]) in
        rank order; `final_out` (device float64 tensor [nq, D]) receives all combined scores."""
        nq = len(query_weights)
        if nq == 1 and final_out is None:                       # the reference's usage (webui.py:586): keep the host side short
            q = query_weights[0]
            n = len(q)
            qt_a = np.fromiter(q.keys(), dtype=np.int32, count=n) if n else np.zeros(1, np.int32)
            qw_a = np.fromiter(q.values(), dtype=np.float64, count=n) if n else np.zeros(1, np.float64)
            qp = np.array((0, n), dtype=np.int32)
            qv = np.ascontiguousarray(query_vectors, dtype=np.float32).reshape(1, -1)
            ids = np.empty((1, k), dtype=np.int32)
            vals = np.empty((1, k), dtype=np.float64)
            _lib.check(self._search_fn(self.bm25._h, self.index._h, qt_a.ctypes.data, qw_a.ctypes.data, qp.ctypes.data, qv.ctypes.data, 1,
                                       self._w_bm25, self._w_sim, k, ids.ctypes.data, vals.ctypes.data, None, _lib.current_stream_ptr()))
            return ids, vals
        # the CSR of the batch without a Python-level loop over terms (that loop was ~1/5 of a 256-query call)
        qp = np.zeros(nq + 1, dtype=np.int32)
        np.cumsum(np.fromiter(map(len, query_weights), dtype=np.int32, count=nq), out=qp[1:])
        nt = int(qp[nq])
        if nt:
            qt_a = np.fromiter(itertools.chain.from_iterable(query_weights), dtype=np.int32, count=nt)                   # dict iteration = keys
            qw_a = np.fromiter(itertools.chain.from_iterable(map(dict.values, query_weights)), dtype=np.float64, count=nt)
        else:
            qt_a = np.zeros(1, dtype=np.int32)
            qw_a = np.zeros(1, dtype=np.float64)
        qv = np.ascontiguousarray(np.atleast_2d(query_vectors), dtype=np.float32)
        ids = np.empty((nq, k), dtype=np.int32)
        vals = np.empty((nq, k), dtype=np.float64)
        _lib.call("hipts_search", self.bm25._h, self.index._h, _lib.ptr(qt_a), _lib.ptr(qw_a), _lib.ptr(qp), _lib.ptr(qv), nq,
                  c_double(BM25_WEIGHT), c_double(DOC2VEC_WEIGHT), k, _lib.ptr(ids), _lib.ptr(vals),
                  _lib.ptr(final_out) if final_out is not None else None, _lib.current_stream_ptr())
        return ids, vals

    def submit_topk(self, query_weights: Sequence[Dict[int, float]], query_vectors: np.ndarray, k: int, slot: int = 0):
        """First half of score_topk for a host that serves a stream of batches: packs and launches the batch into `slot` (0 or 1) and returns
        a ticket without waiting; `collect_topk(ticket)` returns what score_topk would have.  With the two slots used alternately the host
        prepares batch i + 1 while the device runs batch i (hipts_search_submit / hipts_search_collect)."""
        nq = len(query_weights)
        qp = np.zeros(nq + 1, dtype=np.int32)
        np.cumsum(np.fromiter(map(len, query_weights), dtype=np.int32, count=nq), out=qp[1:])
        nt = int(qp[nq])
        if nt:
            qt_a = np.fromiter(itertools.chain.from_iterable(query_weights), dtype=np.int32, count=nt)
            qw_a = np.fromiter(itertools.chain.from_iterable(map(dict.values, query_weights)), dtype=np.float64, count=nt)
        else:
            qt_a = np.zeros(1, dtype=np.int32)
            qw_a = np.zeros(1, dtype=np.float64)
        qv = np.ascontiguousarray(np.atleast_2d(query_vectors), dtype=np.float32)
        _lib.call("hipts_search_submit", self.bm25._h, self.index._h, _lib.ptr(qt_a), _lib.ptr(qw_a), _lib.ptr(qp), _lib.ptr(qv), nq,
                  c_double(BM25_WEIGHT), c_double(DOC2VEC_WEIGHT), k, slot, _lib.current_stream_ptr())
        return (slot, nq, k)

    def collect_topk(self, ticket) -> Tuple[np.ndarray, np.ndarray]:
        slot, nq, k = ticket
        ids = np.empty((nq, k), dtype=np.int32)
        vals = np.empty((nq, k), dtype=np.float64)
        _lib.call("hipts_search_collect", self.bm25._h, slot, _lib.ptr(ids), _lib.ptr(vals))
        return ids, vals

    # ---- webui.py:345-390 -----------------------------------------------------------------------
    def find_similar_documents(self, new_doc: str, topn: int = 50) -> List[Tuple[int, float]]:
        import torch
        self.stats["queries"] += 1
        vec = self.normalize_and_apply_weight_doc2vec(new_doc)                        # :349
        qw, required, exclude = self.parse_bm25_query(new_doc)                        # :354-371
        D = len(self.index)
        final_dev = torch.empty((1, D), dtype=torch.float64, device="cuda:%d" % self.index.device)
        k = min(TOPK_MAX, D)
        ids, vals = self.score_topk([qw], vec[None, :], k, final_out=final_dev)       # :352,374-383
        if self.search_mode == "character oriented":                                  # :386-388
            return self._cfeatures_rerank(ids[0], vals[0], topn, required, exclude)
        return self._doc2vec_rerank(final_dev, ids[0], vals[0], topn)                 # :390

    # ---- webui.py:255-342 -----------------------------------------------------------------------
    def _cfeatures_rerank(self, ids: np.ndarray, vals: np.ndarray, topn: int, required: List[str], exclude: List[str]):
        from .cfeatures import cfeatures_rerank, gen_image_ndarray
        if self.cindex is None:
            raise RuntimeError("character oriented mode needs engine.cindex (a cfeatures.CharacterFeatureIndex with an encoder)")
        D = len(self.index)
        if D <= 10:                                                                   # :336-342
            sims = filter_searched_result([(int(i), float(v)) for i, v in zip(ids[:D], vals[:D])])
            return sims[:min(topn, len(sims))]
        top10 = [(int(i), float(v)) for i, v in zip(ids[:10], vals[:10])]
        feats = []
        for doc_id, _ in top10:                                                       # :292-301
            path = self.image_files_name_tags_arr[doc_id].split(",")[0]
            arr = gen_image_ndarray(path, self.cindex.image_size)
            if arr is None:
                continue
            feats.append(self.cindex.ccip_batch_extract_features([arr])[0])
        if not feats:
            return top10
        return cfeatures_rerank(top10, feats, self.cindex, self.file_tag_index_dict, self.filepath_docid_dict, required, exclude,
                                self.cindex.cosine_diff_threshold)        # own parameter: gen_cfeatures.py:298-299's constant is the metric model's

    # ---- webui.py:189-253 -----------------------------------------------------------------------
    def _doc_tags(self, doc_id: int) -> List[str]:
        return self.image_files_name_tags_arr[doc_id].split(",")[1:]                  # :183 (doc_id+1-1)

    def _rerank_query(self, top10_ids: Sequence[int], top10_scores: Sequence[float]) -> np.ndarray:
        vecs = self.model.infer_vectors([self._doc_tags(int(d)) for d in top10_ids])  # :198-199
        mean = np.average(vecs.astype(np.float64), axis=0, weights=np.asarray(top10_scores, dtype=np.float64))   # :200 (value column)
        if self.compat_rerank:
            # webui.py:200-203 divides the (index, value) pairs by the Frobenius norm of the whole
            # [300,2] array and rounds the indices: every index becomes 0, so the sparse query is
            # sum(values) * e0, which gensim then unit-normalises  [analysis; gensim unavailable].
            q = np.zeros_like(mean)
            s = mean.sum()
            q[0] = 1.0 if s >= 0 else -1.0
            return q
        n = np.linalg.norm(mean)
        return mean / n if n > 0 else mean

    def _ranked_prefix(self, scores_dev, k: int) -> Tuple[np.ndarray, np.ndarray]:
        ids = np.empty((1, k), dtype=np.int32)
        vals = np.empty((1, k), dtype=np.float64)
        _lib.call("hipts_topk", _lib.ptr(scores_dev), 1, c_int64(scores_dev.shape[-1]), k, _lib.ptr(ids), _lib.ptr(vals),
                  _lib.HOST, self.index.device, _lib.current_stream_ptr())
        return ids[0], vals[0]

    def _doc2vec_rerank(self, final_dev, ids: np.ndarray, vals: np.ndarray, topn: int) -> List[Tuple[int, float]]:
        import torch
        D = final_dev.shape[-1]
        if D <= 10:                                                                   # :247-253
            sims = filter_searched_result([(int(i), float(v)) for i, v in zip(ids[:D], vals[:D])])
            return sims[:min(topn, len(sims))]
        top10_ids = [int(i) for i in ids[:10]]
        top10_scores = [float(v) for v in vals[:10]]
        q = self._rerank_query(top10_ids, top10_scores).astype(np.float32)
        rs_dev = torch.empty((1, D), dtype=torch.float32, device=final_dev.device)
        self.index.query(q, out=rs_dev)                                               # :205
        rf_dev = torch.empty_like(final_dev)
        _lib.call("hipts_combine", _lib.ptr(final_dev), _lib.ptr(rs_dev), 1, c_int64(D), c_double(ORIGINAL_SCORE_WEIGHT),
                  c_double(RERANKED_SCORE_WEIGHT), 0, 0, _lib.ptr(rf_dev), self.index.device, _lib.current_stream_ptr())   # :208
        k = min(TOPK_MAX, D)
        rids, rvals = self._ranked_prefix(rf_dev, k)
        last_id, last_val = int(rids[-1]), float(rvals[-1])                          # the last ranked entry as the device holds it
        mx = rvals[0]
        if mx > 0:
            rvals = rvals / mx                                                        # :210-211
        top10_set = set(top10_ids)
        final = [(d, 1.0) for d in top10_ids]                                         # :219-222
        final += [(int(i), float(v)) for i, v in zip(rids, rvals) if int(i) not in top10_set]   # :217,225-237
        # The gap filter (webui.py:63-80) cuts at the SECOND near-tie of the whole ranked list, wherever that is.  Almost always it lies
        # inside the first 1024 ranks; while it does not, the ranking is continued on the device 1024 entries at a time
        # (hipts_topk_after) -- until two cut points are in hand or only -inf scores remain (they produce no cut point: inf - inf is
        # nan).  Round 2 ranked all D scores on the host in that case (27 ms, one query in 20 on the bench corpus).
        n_ranked = k
        while n_ranked < D and last_val > -math.inf and not self._two_cut_points(final):
            self.stats["rank_continuations"] = self.stats.get("rank_continuations", 0) + 1
            kk = min(TOPK_MAX, D - n_ranked)
            mids = np.empty((1, kk), dtype=np.int32)
            mvals = np.empty((1, kk), dtype=np.float64)
            _lib.call("hipts_topk_after", _lib.ptr(rf_dev), c_int64(D), kk, c_double(last_val), c_int64(last_id), _lib.ptr(mids), _lib.ptr(mvals),
                      self.index.device, _lib.current_stream_ptr())
            last_id, last_val = int(mids[0, -1]), float(mvals[0, -1])
            more = mvals[0] / mx if mx > 0 else mvals[0]
            final += [(int(i), float(v)) for i, v in zip(mids[0], more) if int(i) not in top10_set]
            n_ranked += kk
        final = filter_searched_result(final)                                         # :240
        return final[:min(topn, len(final))]

    @staticmethod
    def _two_cut_points(ranked: List[Tuple[int, float]]) -> bool:
        s = np.array([v for _, v in ranked])
        d = s[:-1] - s[1:]
        d = np.where(d == 0, np.inf, d)
        return int((d < DIFF_FILTER_THRESH).sum()) >= 2


_engine: Optional[SearchEngine] = None


def set_engine(engine: SearchEngine):
    global _engine
    _engine = engine


def find_similar_documents(new_doc: str, topn: int = 50) -> List[Tuple[int, float]]:
    """webui.py:345 -- module-level entry point with the reference's signature."""
    if _engine is None:
        raise RuntimeError("no SearchEngine installed: call set_engine(load_engine()) first (webui.py:585 load_model)")
    return _engine.find_similar_documents(new_doc, topn)


def load_engine(device: int = 0, d2v_model: str = "doc2vec_model", compat_rerank: bool = False) -> SearchEngine:
    """webui.py:649-689 load_model(): everything the query function needs, from the files
    genmodel.py writes into the current directory."""
    import pickle
    from .bm25 import load_bm25_index
    tag_file_path = 'tags-wd-tagger_doc2vec_idx.csv'
    with open(tag_file_path, 'r', encoding='utf-8') as f:
        lines = [line.strip() for line in f.readlines()]
    model = Doc2VecInference.load(d2v_model, device)
    index = Similarity.load("doc2vec_index", device)
    dictionary = pickle.load(open("doc2vec_dictionary", "rb"))
    bm25 = load_bm25_index(device)
    return SearchEngine(model, index, dictionary.token2id, bm25, lines, compat_rerank=compat_rerank)
