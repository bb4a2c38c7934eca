"""Doc2Vec PV-DBOW inference oracle -- test infrastructure, see oracle/__init__.py.

ctypes wrapper over oracle/csrc/oracle.c::orc_d2v_infer (PARITY UNPINNED: gensim 4.3.3 is
absent; algorithm restated from doc2vec.py::infer_vector / doc2vec_inner.pyx /
word2vec_inner.pyx; reference call sites genmodel.py:159,169, webui.py:106,185).
"""
import ctypes

import numpy as np

from .search import lib


def exp_table() -> np.ndarray:
    t = np.empty(1000, dtype=np.float32)
    lib().orc_exp_table(t.ctypes.data_as(ctypes.c_void_p))
    return t


def infer(syn1neg: np.ndarray, cum_table: np.ndarray, sample_int, doc_ptr: np.ndarray, words: np.ndarray,
          v0: np.ndarray, seeds: np.ndarray, epochs: int, alpha: float = 0.025, min_alpha: float = 1e-4,
          negative: int = 5, exp_scale: float = 83.0) -> np.ndarray:
    syn1neg = np.ascontiguousarray(syn1neg, dtype=np.float32)
    cum_table = np.ascontiguousarray(cum_table, dtype=np.uint32)
    V, dim = syn1neg.shape
    doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.int32)
    v0 = np.ascontiguousarray(v0, dtype=np.float32)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(doc_ptr) - 1
    out = np.empty((n, dim), dtype=np.float32)
    si = None
    if sample_int is not None:
        si = np.ascontiguousarray(sample_int, dtype=np.uint32)
    f = lib().orc_d2v_infer
    f.restype = None
    f(syn1neg.ctypes.data_as(ctypes.c_void_p), cum_table.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(V),
      si.ctypes.data_as(ctypes.c_void_p) if si is not None else None, ctypes.c_int(dim),
      doc_ptr.ctypes.data_as(ctypes.c_void_p), words.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n),
      v0.ctypes.data_as(ctypes.c_void_p), seeds.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(epochs),
      ctypes.c_float(alpha), ctypes.c_float(min_alpha), ctypes.c_int(negative), ctypes.c_double(exp_scale),
      out.ctypes.data_as(ctypes.c_void_p))
    return out
