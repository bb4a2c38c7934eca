#!/usr/bin/env python3
"""Development aid: time one variant (HIPTS_S1_VARIANT) of the one-query path: C ABI latency + the score kernel's event time."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth, _lib
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
D, V, K, TOPK = 100_000, 10_000, int(os.environ.get("S1_DIM", "300")), 100
ptr, terms = synth.tag_corpus(D, V, seed=42)
rows = synth.index_vectors(D, K, seed=46)
bm = BM25Index(ptr, terms, V, 0)
idx = Similarity("bench", None, K, 0, capacity=D); idx.add_matrix(rows)
qs = [dict(q) for q in synth.queries(64, V, seed=43)]
qv = np.random.default_rng(5).standard_normal((64, K)).astype(np.float32)
lib = _lib.load(); fn = lib.hipts_search
ids = np.empty((1, TOPK), np.int32); vals = np.empty((1, TOPK), np.float64)
def args(i):
    q = qs[i]; qt = np.asarray(list(q.keys()) or [0], np.int32); qw = np.asarray(list(q.values()) or [0.0], np.float64); qp = np.asarray([0, len(q)], np.int32)
    v = np.ascontiguousarray(qv[i:i + 1])
    return (qt, qw, qp, v), (bm._h, idx._h, _lib.ptr(qt), _lib.ptr(qw), _lib.ptr(qp), _lib.ptr(v), 1, ctypes.c_double(0.5), ctypes.c_double(0.5), TOPK, _lib.ptr(ids), _lib.ptr(vals), None, None)
al = [args(i) for i in range(64)]
for keep, a in al[:16]: fn(*a)
t0 = time.perf_counter()
for r in range(8):
    for keep, a in al: fn(*a)
lat = (time.perf_counter() - t0) / 512 * 1e6
_lib.call("hipts_query_profile_enable", bm._h, 1)
for keep, a in al: fn(*a)
out = []
for c in range(5, 9):
    ms, nn, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.call("hipts_query_profile_read", bm._h, c, ctypes.byref(ms), ctypes.byref(nn), ctypes.byref(by))
    out.append(1e3 * ms.value / max(nn.value, 1))
print("variant %s dim %d: C ABI %.1f us/query; events: score %.1f combine %.1f collect %.1f topk %.1f us" % (os.environ.get("HIPTS_S1_VARIANT", "0"), K, lat, *out))
