"""Dense similarity index: host-side mirror of how the reference uses gensim's
`Similarity` / `MatrixSimilarity` (genmodel.py:170-175, gen_cfeatures.py:311-314,360-368,459,
webui.py:205,352,306-309) over libhip_tagsearch's device-resident float32 matrix.

gensim pickles cannot be read or written without gensim; `save`/`load` use an
`.npy` matrix next to a small JSON manifest under the same prefix.
"""
import ctypes
import json
from typing import Iterable, Sequence, Union

import numpy as np

from . import _lib
from ._lib import c_int64, c_void_p


def _as_dense(doc, dim: int) -> np.ndarray:
    """A document is either a dense vector or gensim's sparse [(feature_id, value), ...] form."""
    if isinstance(doc, np.ndarray):
        return doc.astype(np.float32, copy=False)
    if len(doc) > 0 and isinstance(doc[0], (tuple, list)):
        v = np.zeros(dim, dtype=np.float32)
        for i, val in doc:
            v[int(i)] = val
        return v
    return np.asarray(doc, dtype=np.float32)


class Similarity:
    """Similarity(output_prefix, corpus, num_features): rows are stored as given when they are
    ndarrays (genmodel.py:171-173 -> un-normalised document vectors) and L2-normalised when they
    arrive in gensim's sparse list-of-tuples form (gen_cfeatures.py:310-314), which is what gensim's
    index build does [published behaviour, unverifiable here]."""

    def __init__(self, output_prefix: str, corpus: Iterable, num_features: int, device: int = 0, capacity: int = 0):
        self.output_prefix = output_prefix
        self.num_features = int(num_features)
        self.device = device
        self._h = c_void_p()
        _lib.call("hipts_index_create", self.num_features, c_int64(capacity), device, ctypes.byref(self._h))
        if corpus is not None:
            self.add_documents(corpus)

    # --- building ------------------------------------------------------------------------
    def add_documents(self, corpus: Iterable):
        rows = []
        for doc in corpus:
            sparse = (not isinstance(doc, np.ndarray)) and len(doc) > 0 and isinstance(doc[0], (tuple, list))
            v = _as_dense(doc, self.num_features)
            if sparse:
                n = np.float32(np.sqrt(np.sum(v.astype(np.float32) ** 2)))
                if n > 0:
                    v = v / n
            rows.append(v)
        if rows:
            self.add_matrix(np.stack(rows).astype(np.float32))

    def add_matrix(self, rows):
        """Append a float32 [n, num_features] block (numpy array or torch tensor, host or device)."""
        if isinstance(rows, np.ndarray):
            rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert tuple(rows.shape)[1] == self.num_features
        _lib.call("hipts_index_add", self._h, _lib.ptr(rows), c_int64(int(rows.shape[0])), _lib.memspace_of(rows))

    # --- queries -------------------------------------------------------------------------
    def __len__(self) -> int:
        n = c_int64()
        _lib.call("hipts_index_len", self._h, ctypes.byref(n))
        return n.value

    def vector_by_id(self, docpos: int) -> np.ndarray:
        out = np.empty(self.num_features, dtype=np.float32)
        _lib.call("hipts_index_vector_by_id", self._h, c_int64(docpos), _lib.ptr(out))
        return out

    def query(self, queries: np.ndarray, out=None) -> np.ndarray:
        """scores float32 [nq, len]: rows . query, fused k-ordered float32 chain (exact-f32 MFMA)."""
        queries = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        nq = queries.shape[0]
        if out is None:
            out = np.empty((nq, len(self)), dtype=np.float32)
        _lib.call("hipts_index_query", self._h, _lib.ptr(queries), _lib.HOST, nq, _lib.ptr(out), _lib.memspace_of(out),
                  _lib.current_stream_ptr())
        return out

    def __getitem__(self, query: Union[np.ndarray, Sequence]) -> np.ndarray:
        """index[vec] as webui.py:205,352 calls it: `vec` in gensim's sparse form is unit-normalised
        by gensim before the product [published behaviour]; dense vectors are used as given."""
        sparse = (not isinstance(query, np.ndarray)) and len(query) > 0 and isinstance(query[0], (tuple, list))
        v = _as_dense(query, self.num_features)
        if sparse:
            n = np.float32(np.sqrt(np.sum(v ** 2)))
            if n > 0:
                v = v / n
        return self.query(v)[0]

    # --- persistence ---------------------------------------------------------------------
    def matrix(self, first: int = 0, nrows: int = None) -> np.ndarray:
        """Rows [first, first + nrows) as one host array: a single device-to-host copy (hipts_index_export)."""
        n = len(self)
        nrows = n - first if nrows is None else nrows
        out = np.empty((nrows, self.num_features), dtype=np.float32)
        _lib.call("hipts_index_export", self._h, c_int64(first), c_int64(nrows), _lib.ptr(out))
        return out

    def save(self, fname: str = None):
        """Manifest `fname` (JSON) + `fname.npy`.  NOT gensim's pickle layout (unreadable / unwritable without gensim,
        SURVEY.md T1): index files are not interchangeable with the reference's in either direction.
        Written to temporary names and renamed, so a crash leaves the previous pair intact."""
        import os
        fname = fname or self.output_prefix
        with open(fname + ".npy.tmp", "wb") as f:
            np.save(f, self.matrix())
        with open(fname + ".tmp", "w") as f:
            json.dump({"format": "hiptagsearch-dense-index-v1", "num_features": self.num_features, "rows": len(self)}, f)
        os.replace(fname + ".npy.tmp", fname + ".npy")
        os.replace(fname + ".tmp", fname)

    @classmethod
    def load(cls, fname: str, device: int = 0, mmap=None) -> "Similarity":
        meta = json.load(open(fname))
        rows = np.load(fname + ".npy", mmap_mode="r")
        if rows.shape != (meta["rows"], meta["num_features"]):
            raise ValueError("%s.npy is %r but the manifest says %d x %d" % (fname, rows.shape, meta["rows"], meta["num_features"]))
        idx = cls(fname, None, meta["num_features"], device, capacity=meta["rows"])
        step = max(1, (256 << 20) // (4 * meta["num_features"]))            # stream large matrices in 256 MB blocks
        for s0 in range(0, meta["rows"], step):
            idx.add_matrix(np.ascontiguousarray(rows[s0:s0 + step]))
        return idx

    def close(self):
        if self._h:
            _lib.call("hipts_index_destroy", self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MatrixSimilarity = Similarity   # webui.py:7,28 loads the same files through this name
