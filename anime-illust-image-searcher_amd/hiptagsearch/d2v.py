"""Doc2Vec PV-DBOW inference: host-side mirror of `gensim.models.Doc2Vec.infer_vector` as the
reference uses it (genmodel.py:169; webui.py:106,185) over libhip_tagsearch's wave-per-document
kernel (csrc/d2v.hip).  Training (genmodel.py:159-162) is out of scope; a model is the frozen
arrays inference consumes: syn1neg, cum_table, sample_int and the vocabulary.
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
    """Frozen PV-DBOW model on the device."""

    def __init__(self, syn1neg: np.ndarray, cum_table: np.ndarray, sample_int: Optional[np.ndarray],
                 key_to_index: Dict[str, int], epochs: int = 100, alpha: float = 0.025, min_alpha: float = 1e-4,
                 negative: int = 5, exp_scale: float = 83.0, seed: int = 1, device: int = 0):
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
