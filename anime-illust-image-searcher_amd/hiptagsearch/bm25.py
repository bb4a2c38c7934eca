"""BM25 host-side mirror of the reference interface over the C ABI.

  gen_and_save_bm25_index(corpus, dictionary)   genmodel.py:51-99   (same name, same five pickles)
  compute_bm25_scores(query_weights=...)         webui.py:119-172
The statistics pass and the scoring kernel live in libhip_tagsearch.so (csrc/query.hip).
"""
import ctypes
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import c_double, c_int32, c_int64, c_void_p


def _csr_from_tokens(corpus: Sequence[Sequence[str]], token2id: Dict[str, int]):
    ptr = np.zeros(len(corpus) + 1, dtype=np.int64)
    ids: List[int] = []
    for i, tags in enumerate(corpus):
        ids.extend(token2id.get(t, -1) for t in tags)
        ptr[i + 1] = len(ids)
    return ptr, np.asarray(ids, dtype=np.int32)


class BM25Index:
    """Device-resident BM25 statistics (document-major CSR of (term, tf), doc lengths, idf)."""

    def __init__(self, doc_ptr: np.ndarray, term_ids: np.ndarray, vocab: int, device: int = 0,
                 numpy_idf: bool = True):
        doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
        term_ids = np.ascontiguousarray(term_ids, dtype=np.int32)
        self._h = c_void_p()
        self.device = device
        _lib.call("hipts_bm25_build", _lib.ptr(doc_ptr), _lib.ptr(term_ids), c_int64(len(doc_ptr) - 1),
                  c_int32(vocab), device, ctypes.byref(self._h))
        D, nnz, V, avgdl = c_int64(), c_int64(), c_int32(), c_double()
        _lib.call("hipts_bm25_info", self._h, ctypes.byref(D), ctypes.byref(nnz), ctypes.byref(V), ctypes.byref(avgdl))
        self.D, self.nnz, self.vocab, self.avgdl = D.value, nnz.value, V.value, np.float64(avgdl.value)
        if numpy_idf:
            # genmodel.py:80-82 evaluates np.log on Python scalars; libm's log differs from numpy's
            # in the last bit for ~0.3 % of arguments, so the table is recomputed with the
            # reference's exact expression and installed (keeps the bm25_idf pickle bit-identical).
            df = self.export()["df"]
            idf = np.zeros(self.vocab, dtype=np.float64)
            for t in np.nonzero(df)[0]:
                idf[t] = np.log(1 + (self.D - int(df[t]) + 0.5) / (int(df[t]) + 0.5))
            self.set_idf(idf)

    @classmethod
    def from_tokens(cls, corpus: Sequence[Sequence[str]], token2id: Dict[str, int], device: int = 0):
        ptr, ids = _csr_from_tokens(corpus, token2id)
        return cls(ptr, ids, (max(token2id.values()) + 1) if token2id else 0, device)

    def set_idf(self, idf: np.ndarray):
        idf = np.ascontiguousarray(idf, dtype=np.float64)
        assert idf.shape == (self.vocab,)
        _lib.call("hipts_bm25_set_idf", self._h, _lib.ptr(idf))

    def export(self) -> Dict[str, np.ndarray]:
        out = {"csr_ptr": np.empty(self.D + 1, np.int64), "csr_term": np.empty(self.nnz, np.int32),
               "csr_tf": np.empty(self.nnz, np.int32), "doc_len": np.empty(self.D, np.int64),
               "df": np.empty(self.vocab, np.int64), "idf": np.empty(self.vocab, np.float64)}
        _lib.call("hipts_bm25_export", self._h, _lib.ptr(out["csr_ptr"]), _lib.ptr(out["csr_term"]),
                  _lib.ptr(out["csr_tf"]), _lib.ptr(out["doc_len"]), _lib.ptr(out["df"]), _lib.ptr(out["idf"]))
        return out

    def reference_objects(self):
        """The five objects genmodel.py:84-97 pickles, with the reference's Python types."""
        e = self.export()
        corpus = []
        ptr, term, tf = e["csr_ptr"], e["csr_term"], e["csr_tf"]
        for d in range(self.D):
            s, t = int(ptr[d]), int(ptr[d + 1])
            corpus.append({int(a): int(b) for a, b in zip(term[s:t], tf[s:t])})
        # dict order of bm25_idf = first time a term is seen walking documents in order (:72-82)
        _, first = np.unique(term, return_index=True)
        order = term[np.sort(first)]
        idf = {int(t): np.float64(e["idf"][t]) for t in order}
        return corpus, idf, np.float64(self.avgdl), int(self.D), e["doc_len"]

    def score(self, query_weights_list: Sequence[Dict[int, float]], out=None) -> np.ndarray:
        """scores float64 [nq, D] for a list of {term_id: weight} dicts (dict order is kept)."""
        nq = len(query_weights_list)
        qp = np.zeros(nq + 1, dtype=np.int32)
        qt: List[int] = []
        qw: List[float] = []
        for i, q in enumerate(query_weights_list):
            for t, w in q.items():
                qt.append(int(t))
                qw.append(float(w))
            qp[i + 1] = len(qt)
        qt_a = np.asarray(qt if qt else [0], dtype=np.int32)
        qw_a = np.asarray(qw if qw else [0.0], dtype=np.float64)
        if out is None:
            out = np.empty((nq, self.D), dtype=np.float64)
        _lib.call("hipts_bm25_score", self._h, _lib.ptr(qt_a), _lib.ptr(qw_a), _lib.ptr(qp), nq, _lib.ptr(out),
                  _lib.memspace_of(out), _lib.current_stream_ptr())
        return out

    def close(self):
        if self._h:
            _lib.call("hipts_bm25_destroy", self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gen_and_save_bm25_index(corpus: List[List[str]], dictionary, device: int = 0) -> BM25Index:
    """Same name, arguments and output files as genmodel.py:51."""
    idx = BM25Index.from_tokens(corpus, dictionary.token2id, device)
    bm25_corpus, bm25_idf, bm25_avgdl, bm25_D, bm25_doc_lengths = idx.reference_objects()
    for name, obj in (("bm25_corpus", bm25_corpus), ("bm25_idf", bm25_idf), ("bm25_avgdl", bm25_avgdl),
                      ("bm25_D", bm25_D), ("bm25_doc_lengths", bm25_doc_lengths)):
        with open(name, "wb") as f:
            pickle.dump(obj, f)
    print("BM25 index generated")
    return idx


def load_bm25_index(device: int = 0) -> BM25Index:
    """Rebuild the device index from the reference's five pickles (webui.py:676-680)."""
    corpus = pickle.load(open("bm25_corpus", "rb"))
    idf = pickle.load(open("bm25_idf", "rb"))
    ptr = np.zeros(len(corpus) + 1, dtype=np.int64)
    ids: List[int] = []
    for i, d in enumerate(corpus):
        for t, f in d.items():
            ids.extend([t] * f)
        ptr[i + 1] = len(ids)
    vocab = (max(idf.keys()) + 1) if idf else 0
    idx = BM25Index(ptr, np.asarray(ids, dtype=np.int32), vocab, device, numpy_idf=False)
    table = np.zeros(vocab, dtype=np.float64)
    for t, v in idf.items():
        table[t] = v
    idx.set_idf(table)
    return idx


def compute_bm25_scores(index: BM25Index, query_terms: Sequence[str] = (), query_weights: Optional[Dict[int, float]] = None,
                        dictionary=None) -> np.ndarray:
    """webui.py:119 with the index passed explicitly instead of through module globals."""
    if query_weights is None:
        ids = [dictionary.token2id[t] for t in query_terms if t in dictionary.token2id]   # webui.py:133-134
        query_weights = {i: 1.0 for i in ids}
    return index.score([query_weights])[0]
