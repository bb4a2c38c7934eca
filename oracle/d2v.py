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


def infer_dm(syn1neg: np.ndarray, word_vectors: np.ndarray, cum_table: np.ndarray, sample_int, doc_ptr: np.ndarray, words: np.ndarray,
             v0: np.ndarray, seeds: np.ndarray, epochs: int, alpha: float = 0.025, min_alpha: float = 1e-4, negative: int = 5,
             exp_scale: float = 83.0, window: int = 5, dm_mean: int = 1) -> np.ndarray:
    """PV-DM inference (orc_d2v_infer_dm: doc2vec_inner.pyx::train_document_dm with frozen word vectors and hidden layer)."""
    syn1neg = np.ascontiguousarray(syn1neg, dtype=np.float32)
    word_vectors = np.ascontiguousarray(word_vectors, dtype=np.float32)
    cum_table = np.ascontiguousarray(cum_table, dtype=np.uint32)
    V, dim = syn1neg.shape
    assert word_vectors.shape == (V, dim)
    doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.int32)
    v0 = np.ascontiguousarray(v0, dtype=np.float32)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
    n = len(doc_ptr) - 1
    out = np.empty((n, dim), dtype=np.float32)
    si = None if sample_int is None else np.ascontiguousarray(sample_int, dtype=np.uint32)
    f = lib().orc_d2v_infer_dm
    f.restype = None
    P = ctypes.c_void_p
    f(syn1neg.ctypes.data_as(P), word_vectors.ctypes.data_as(P), cum_table.ctypes.data_as(P), ctypes.c_int64(V),
      si.ctypes.data_as(P) if si is not None else None, ctypes.c_int(dim), doc_ptr.ctypes.data_as(P), words.ctypes.data_as(P),
      ctypes.c_int64(n), v0.ctypes.data_as(P), seeds.ctypes.data_as(P), ctypes.c_int(epochs), ctypes.c_float(alpha),
      ctypes.c_float(min_alpha), ctypes.c_int(negative), ctypes.c_double(exp_scale), ctypes.c_int(window), ctypes.c_int(dm_mean),
      out.ctypes.data_as(P))
    return out


def build_vocab(docs, sample: float = 1e-3, ns_exponent: float = 0.75):
    """gensim Doc2Vec.build_vocab with min_count=1 as genmodel.py:159-160 calls it [published algorithm, PARITY UNPINNED]:
    vocabulary sorted by descending count (ties: first occurrence first), `sample_int` from the sub-sampling formula of
    word2vec.py::prepare_vocab, `cum_table` from make_cum_table (domain 2^31 - 1).  Returns (key_to_index, counts int64,
    cum_table uint32, sample_int uint32)."""
    first, counts = {}, {}
    for d in docs:
        for t in d:
            if t not in counts:
                first[t] = len(first)
                counts[t] = 0
            counts[t] += 1
    vocab = sorted(counts, key=lambda t: (-counts[t], first[t]))
    cnt = np.array([counts[t] for t in vocab], dtype=np.int64)
    retain_total = int(cnt.sum())
    threshold_count = sample * retain_total if sample < 1.0 else int(sample * (3 + np.sqrt(5)) / 2)
    sample_int = np.empty(len(vocab), dtype=np.uint32)
    for i, v in enumerate(cnt):
        p = (np.sqrt(v / threshold_count) + 1) * (threshold_count / v)
        sample_int[i] = np.uint32(min(p, 1.0) * (2 ** 32 - 1))
    domain = 2 ** 31 - 1
    pw = cnt.astype(np.float64) ** ns_exponent
    total = float(pw.sum())
    cum_table = np.zeros(len(vocab), dtype=np.uint32)
    cumulative = 0.0
    for i in range(len(vocab)):
        cumulative += pw[i]
        cum_table[i] = round(cumulative / total * domain)
    assert cum_table[-1] == domain
    return {t: i for i, t in enumerate(vocab)}, cnt, cum_table, sample_int


def init_doc_vectors(ndocs: int, dim: int, seed: int = 1) -> np.ndarray:
    """gensim 4 KeyedVectors.resize_vectors(seed) -> prep_vectors: uniform in [-1/dim, 1/dim) from default_rng(seed) [published]."""
    rng = np.random.default_rng(seed)
    v = rng.random((ndocs, dim), dtype=np.float32)
    v *= np.float32(2.0)
    v -= np.float32(1.0)
    v /= np.float32(dim)
    return v


def train(syn1neg: np.ndarray, doc_vectors: np.ndarray, cum_table: np.ndarray, sample_int, doc_ptr: np.ndarray, words: np.ndarray,
          epochs: int, alpha: float = 0.025, min_alpha: float = 1e-4, negative: int = 5, exp_scale: float = 83.0, seed: int = 1,
          batch_words: int = 10000):
    """In-place sequential training (orc_d2v_train).  Returns (syn1neg, doc_vectors)."""
    assert syn1neg.dtype == np.float32 and doc_vectors.dtype == np.float32 and syn1neg.flags["C_CONTIGUOUS"] and doc_vectors.flags["C_CONTIGUOUS"]
    cum_table = np.ascontiguousarray(cum_table, dtype=np.uint32)
    V, dim = syn1neg.shape
    doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.int32)
    si = None if sample_int is None else np.ascontiguousarray(sample_int, dtype=np.uint32)
    f = lib().orc_d2v_train
    f.restype = None
    vp = ctypes.c_void_p
    f(syn1neg.ctypes.data_as(vp), doc_vectors.ctypes.data_as(vp), cum_table.ctypes.data_as(vp), ctypes.c_int64(V),
      si.ctypes.data_as(vp) if si is not None else None, ctypes.c_int(dim), doc_ptr.ctypes.data_as(vp), words.ctypes.data_as(vp),
      ctypes.c_int64(len(doc_ptr) - 1), ctypes.c_int(epochs), ctypes.c_float(alpha), ctypes.c_float(min_alpha), ctypes.c_int(negative),
      ctypes.c_double(exp_scale), ctypes.c_uint64(seed), ctypes.c_int(batch_words))
    return syn1neg, doc_vectors
