"""Doc2Vec PV-DBOW (and, as a flag, PV-DM) inference: host-side mirror of `gensim.models.Doc2Vec.infer_vector` as the
reference uses it (genmodel.py:169; webui.py:106,185) over libhip_tagsearch's wave-per-document
kernel (csrc/d2v.hip), and of `Doc2Vec(...)` / `build_vocab` / `train` (genmodel.py:159-162) over
hipts_d2v_train.  A trained model is the arrays inference consumes: syn1neg, cum_table, sample_int
and the vocabulary.
"""
import ctypes
import json
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import c_double, c_float, c_int64, c_void_p


def pseudorandom_weak_vector(size: int, seed_string: str) -> np.ndarray:
    """gensim.utils.pseudorandom_weak_vector with a process-independent hash: gensim seeds SFC64
    with Python's hash(seed_string) (randomised per process unless PYTHONHASHSEED is set); crc32 of
    the UTF-8 bytes is used here so that repeated runs agree."""
    seed = zlib.crc32(seed_string.encode("utf-8")) & 0xFFFFFFFF
    once = np.random.Generator(np.random.SFC64(seed))
    return ((once.random(size).astype(np.float32) - np.float32(0.5)) / np.float32(size)).astype(np.float32)


class Doc2VecInference:
    """Frozen Doc2Vec model on the device: PV-DBOW (dm=0, the reference's configuration, genmodel.py:159) or -- with
    `word_vectors` -- PV-DM (dm=1, non-concatenative; dm_mean 0: sum, 1: mean), the form BASELINE.json's north_star names.
    A dm=1 model is gensim's {wv.vectors, syn1neg, cum_table, sample_int, window}; this package trains DBOW only, so such a model
    comes from outside (or from synthetic arrays in the tests)."""

    def __init__(self, syn1neg: np.ndarray, cum_table: np.ndarray, sample_int: Optional[np.ndarray],
                 key_to_index: Dict[str, int], epochs: int = 100, alpha: float = 0.025, min_alpha: float = 1e-4,
                 negative: int = 5, exp_scale: float = 83.0, seed: int = 1, device: int = 0,
                 dm: int = 0, word_vectors: Optional[np.ndarray] = None, window: int = 5, dm_mean: int = 0):
        self.syn1neg = np.ascontiguousarray(syn1neg, dtype=np.float32)
        self.cum_table = np.ascontiguousarray(cum_table, dtype=np.uint32)
        self.sample_int = None if sample_int is None else np.ascontiguousarray(sample_int, dtype=np.uint32)
        self.key_to_index = key_to_index
        self.vector_size = int(self.syn1neg.shape[1])
        self.epochs, self.alpha, self.min_alpha = int(epochs), float(alpha), float(min_alpha)
        self.negative, self.exp_scale, self.seed, self.device = int(negative), float(exp_scale), int(seed), device
        self._calls = 0
        self._h = c_void_p()
        _lib.call("hipts_d2v_create", _lib.ptr(self.syn1neg), _lib.ptr(self.cum_table),
                  _lib.ptr(self.sample_int) if self.sample_int is not None else None,
                  c_int64(self.syn1neg.shape[0]), self.vector_size, self.negative, c_double(self.exp_scale), device,
                  ctypes.byref(self._h))
        self.dm, self.window, self.dm_mean = int(dm), int(window), int(dm_mean)
        self.word_vectors = None
        if self.dm:
            if word_vectors is None:
                raise ValueError("dm=1 needs the model's word vectors (wv.vectors)")
            self.word_vectors = np.ascontiguousarray(word_vectors, dtype=np.float32)
            if self.word_vectors.shape != self.syn1neg.shape:
                raise ValueError("word_vectors must be [vocab][vector_size] like syn1neg")
            _lib.call("hipts_d2v_set_word_vectors", self._h, _lib.ptr(self.word_vectors))

    # -- raw batch interface (explicit start vectors and seeds) ------------------------------
    def infer_batch(self, doc_ptr: np.ndarray, words: np.ndarray, v0: np.ndarray, seeds: np.ndarray,
                    epochs: Optional[int] = None, out=None) -> np.ndarray:
        doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.int64)
        words = np.ascontiguousarray(words, dtype=np.int32)
        v0 = np.ascontiguousarray(v0, dtype=np.float32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        n = len(doc_ptr) - 1
        if out is None:
            out = np.empty((n, self.vector_size), dtype=np.float32)
        if self.dm:
            _lib.call("hipts_d2v_infer_dm", self._h, _lib.ptr(doc_ptr), _lib.ptr(words if len(words) else np.zeros(1, np.int32)),
                      c_int64(n), _lib.ptr(v0), _lib.ptr(seeds), int(epochs or self.epochs), c_float(self.alpha),
                      c_float(self.min_alpha), self.window, self.dm_mean, _lib.ptr(out), _lib.memspace_of(out), _lib.current_stream_ptr())
            return out
        _lib.call("hipts_d2v_infer", self._h, _lib.ptr(doc_ptr), _lib.ptr(words if len(words) else np.zeros(1, np.int32)),
                  c_int64(n), _lib.ptr(v0), _lib.ptr(seeds), int(epochs or self.epochs), c_float(self.alpha),
                  c_float(self.min_alpha), _lib.ptr(out), _lib.memspace_of(out), _lib.current_stream_ptr())
        return out

    # -- gensim-shaped interface --------------------------------------------------------------
    def _seed_for(self, words: Sequence[str]) -> int:
        """gensim draws each epoch's LCG state from the model's mutable RandomState, so its result
        depends on call history; here it is a pure function of (model seed, words)."""
        return (zlib.crc32((" ".join(words)).encode("utf-8")) * 0x9E3779B1 + self.seed) & 0x7FFFFFFFFFFFFFFF

    def infer_vectors(self, docs: Sequence[Sequence[str]], epochs: Optional[int] = None) -> np.ndarray:
        ptr = np.zeros(len(docs) + 1, dtype=np.int64)
        ids: List[int] = []
        for i, d in enumerate(docs):
            ids.extend(self.key_to_index.get(t, -1) for t in d)
            ptr[i + 1] = len(ids)
        v0 = np.stack([pseudorandom_weak_vector(self.vector_size, " ".join(d)) for d in docs])
        seeds = np.asarray([self._seed_for(d) for d in docs], dtype=np.uint64)
        return self.infer_batch(ptr, np.asarray(ids, dtype=np.int32), v0, seeds, epochs)

    def infer_vector(self, doc_words: Sequence[str], alpha=None, min_alpha=None, epochs=None) -> np.ndarray:
        """Doc2Vec.infer_vector(doc_words) -> float32[vector_size]   (genmodel.py:169)."""
        return self.infer_vectors([list(doc_words)], epochs)[0]

    # -- persistence (gensim's pickle is unreadable without gensim) ------------------------------
    def save(self, fname: str):
        np.savez(fname + ".npz", syn1neg=self.syn1neg, cum_table=self.cum_table,
                 sample_int=self.sample_int if self.sample_int is not None else np.zeros(0, np.uint32))
        json.dump({"format": "hiptagsearch-d2v-v1", "key_to_index": self.key_to_index, "epochs": self.epochs,
                   "alpha": self.alpha, "min_alpha": self.min_alpha, "negative": self.negative,
                   "exp_scale": self.exp_scale, "seed": self.seed}, open(fname, "w"))

    @classmethod
    def load(cls, fname: str, device: int = 0) -> "Doc2VecInference":
        meta = json.load(open(fname))
        arr = np.load(fname + ".npz")
        si = arr["sample_int"]
        return cls(arr["syn1neg"], arr["cum_table"], si if len(si) else None, meta["key_to_index"], meta["epochs"],
                   meta["alpha"], meta["min_alpha"], meta["negative"], meta["exp_scale"], meta["seed"], device)

    def close(self):
        if self._h:
            _lib.call("hipts_d2v_destroy", self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Doc2Vec:
    """gensim.models.Doc2Vec as genmodel.py:159-162 uses it:

        doc2vec_model = Doc2Vec(vector_size=300, window=50, min_count=1, workers=1, dm=0)
        doc2vec_model.build_vocab(tagged_docs)
        doc2vec_model.train(tagged_docs, total_examples=doc2vec_model.corpus_count, epochs=100)
        doc2vec_model.save("doc2vec_model");  doc2vec_model.infer_vector(doc)

    Only PV-DBOW (dm=0) with negative sampling is built -- the configuration the reference runs.  `window` is accepted and
    unused (DBOW without dbow_words has no context window).  Documents are lists of words or objects with a `.words` attribute
    (gensim's TaggedDocument).  gensim's defaults are kept: negative=5, sample=1e-3, alpha=0.025, min_alpha=1e-4,
    ns_exponent=0.75, seed=1, epochs=10 unless train(epochs=...) says otherwise.
    `workers=1` (the reference's setting) selects the sequential, reproducible schedule when the corpus is small enough for one
    wavefront (<= sequential_max_words word occurrences per epoch); otherwise -- and always with workers > 1 -- every document of
    an epoch trains concurrently (lock-free hidden-layer updates, as gensim's own worker threads)."""

    def __init__(self, vector_size: int = 100, window: int = 5, min_count: int = 1, workers: int = 1, dm: int = 0, negative: int = 5,
                 sample: float = 1e-3, alpha: float = 0.025, min_alpha: float = 1e-4, ns_exponent: float = 0.75, seed: int = 1,
                 epochs: int = 10, batch_words: int = 10000, exp_scale: float = 83.0, device: int = 0, sequential_max_words: int = 200_000):
        if dm != 0:
            raise NotImplementedError("training is PV-DBOW only (dm=0, the reference's configuration, genmodel.py:159); PV-DM INFERENCE of a "
                                      "model trained elsewhere: Doc2VecInference(..., dm=1, word_vectors=..., window=..., dm_mean=...)")
        self.vector_size, self.window, self.min_count, self.workers = int(vector_size), window, int(min_count), int(workers)
        self.negative, self.sample, self.alpha, self.min_alpha = int(negative), float(sample), float(alpha), float(min_alpha)
        self.ns_exponent, self.seed, self.epochs, self.batch_words = float(ns_exponent), int(seed), int(epochs), int(batch_words)
        self.exp_scale, self.device, self.sequential_max_words = float(exp_scale), device, int(sequential_max_words)
        self.key_to_index: Dict[str, int] = {}
        self.counts = self.cum_table = self.sample_int = self.syn1neg = self.doc_vectors = None
        self.corpus_count = 0
        self._inference: Optional[Doc2VecInference] = None
        self.last_mode = None

    @staticmethod
    def _words(doc):
        return doc.words if hasattr(doc, "words") else doc

    # ---- word2vec.py::scan_vocab / prepare_vocab / make_cum_table (min_count, sample, ns_exponent) --------------------------
    def build_vocab(self, corpus_iterable) -> None:
        first, counts = {}, {}
        n = 0
        for doc in corpus_iterable:
            n += 1
            for t in self._words(doc):
                if t not in counts:
                    first[t] = len(first)
                    counts[t] = 0
                counts[t] += 1
        self.corpus_count = n
        vocab = sorted((t for t in counts if counts[t] >= self.min_count), key=lambda t: (-counts[t], first[t]))   # descending frequency
        cnt = np.array([counts[t] for t in vocab], dtype=np.int64)
        retain_total = int(cnt.sum())
        threshold_count = self.sample * retain_total if self.sample < 1.0 else int(self.sample * (3 + np.sqrt(5)) / 2)
        if self.sample > 0:
            si = np.empty(len(vocab), dtype=np.uint32)
            for i, v in enumerate(cnt):
                p = (np.sqrt(v / threshold_count) + 1) * (threshold_count / v)
                si[i] = np.uint32(min(p, 1.0) * (2 ** 32 - 1))
            self.sample_int = si
        else:
            self.sample_int = None
        domain = 2 ** 31 - 1
        pw = cnt.astype(np.float64) ** self.ns_exponent
        total = float(pw.sum())
        cum = np.zeros(len(vocab), dtype=np.uint32)
        cumulative = 0.0
        for i in range(len(vocab)):
            cumulative += pw[i]
            cum[i] = round(cumulative / total * domain)
        self.key_to_index = {t: i for i, t in enumerate(vocab)}
        self.counts, self.cum_table = cnt, cum
        # init_weights: hidden layer zero, document vectors uniform in +-1/vector_size from default_rng(seed)
        self.syn1neg = np.zeros((len(vocab), self.vector_size), dtype=np.float32)
        rng = np.random.default_rng(self.seed)
        dv = rng.random((n, self.vector_size), dtype=np.float32)
        dv *= np.float32(2.0)
        dv -= np.float32(1.0)
        dv /= np.float32(self.vector_size)
        self.doc_vectors = dv

    def _csr(self, corpus_iterable):
        ptr, ids = [0], []
        for doc in corpus_iterable:
            ids.extend(self.key_to_index.get(t, -1) for t in self._words(doc))
            ptr.append(len(ids))
        return np.asarray(ptr, dtype=np.int64), np.asarray(ids if ids else [0], dtype=np.int32)

    def train(self, corpus_iterable, total_examples: Optional[int] = None, epochs: Optional[int] = None, mode: Optional[str] = None) -> None:
        """mode: None = by `workers` (see the class docstring), 'sequential' or 'parallel'."""
        if self.syn1neg is None:
            raise RuntimeError("you must first build vocabulary before training the model")       # gensim's message
        epochs = int(epochs if epochs is not None else self.epochs)
        self.epochs = epochs                                                                      # infer_vector's default afterwards
        ptr, ids = self._csr(corpus_iterable)
        n = len(ptr) - 1
        if total_examples is not None and total_examples != n:
            raise ValueError("total_examples=%d but the corpus holds %d documents" % (total_examples, n))
        if n != len(self.doc_vectors):
            raise ValueError("train() needs the corpus build_vocab() saw (%d documents, got %d)" % (len(self.doc_vectors), n))
        if mode is None:
            mode = "sequential" if (self.workers <= 1 and int(ptr[-1]) <= self.sequential_max_words) else "parallel"
        self.last_mode = mode
        _lib.call("hipts_d2v_train", _lib.ptr(self.cum_table), _lib.ptr(self.sample_int) if self.sample_int is not None else None,
                  c_int64(len(self.cum_table)), self.vector_size, self.negative, c_double(self.exp_scale), _lib.ptr(ptr), _lib.ptr(ids), c_int64(n),
                  _lib.ptr(self.doc_vectors), _lib.ptr(self.syn1neg), epochs, c_float(self.alpha), c_float(self.min_alpha),
                  ctypes.c_uint64(self.seed), self.batch_words, 0 if mode == "sequential" else 1, self.device, _lib.current_stream_ptr())
        self._inference = None

    # ---- inference / persistence through Doc2VecInference -------------------------------------------------------------------
    def inference(self) -> Doc2VecInference:
        if self._inference is None:
            self._inference = Doc2VecInference(self.syn1neg, self.cum_table, self.sample_int, self.key_to_index, self.epochs, self.alpha,
                                               self.min_alpha, self.negative, self.exp_scale, self.seed, self.device)
        return self._inference

    def infer_vector(self, doc_words: Sequence[str], alpha=None, min_alpha=None, epochs=None) -> np.ndarray:
        return self.inference().infer_vector(doc_words, alpha, min_alpha, epochs)

    def save(self, fname: str):
        self.inference().save(fname)
        np.save(fname + ".dv.npy", self.doc_vectors)
