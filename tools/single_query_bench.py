#!/usr/bin/env python3
"""Development aid: one query at a time through the fused search path (the webui's usage), for rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
from hiptagsearch.search import SearchEngine
D, V, K, TOPK = 100_000, 10_000, 300, 100
ptr, terms = synth.tag_corpus(D, V, seed=42)
rows = synth.index_vectors(D, K, seed=46)
bm = BM25Index(ptr, terms, V, 0)
idx = Similarity("bench", None, K, 0, capacity=D)
idx.add_matrix(rows)
eng = SearchEngine(None, idx, {}, bm, [])
qs = [dict(q) for q in synth.queries(256, V, seed=43)]
rng = np.random.default_rng(5)
qv = rng.standard_normal((256, K)); qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
for i in range(8): eng.score_topk(qs[i:i + 1], qv[i:i + 1], TOPK)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 200
for i in range(n): eng.score_topk(qs[i:i + 1], qv[i:i + 1], TOPK)
torch.cuda.synchronize()
print("single query: %.1f us / query" % ((time.perf_counter() - t0) / n * 1e6))
# per-kernel HIP-event times of the same loop (hipts_query_profile_*)
import ctypes
from hiptagsearch import _lib
_lib.call("hipts_query_profile_enable", bm._h, 1)
for i in range(64): eng.score_topk(qs[i:i + 1], qv[i:i + 1], TOPK)
for c in range(9):
    ms, nn, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.call("hipts_query_profile_read", bm._h, c, ctypes.byref(ms), ctypes.byref(nn), ctypes.byref(by))
    name = ctypes.create_string_buffer(64); _lib.call("hipts_query_profile_name", c, name, 64)
    if nn.value: print("%-28s n=%3d avg %8.1f us  %7.1f GB/s" % (name.value.decode(), nn.value, 1e3 * ms.value / nn.value, by.value / (ms.value * 1e6)))
_lib.call("hipts_query_profile_enable", bm._h, 0)
# the C ABI alone (no Python marshalling per call): the same query repeated
import numpy as np
plain = [i for i, q in enumerate(qs) if all(0 < w < 1000 for w in q.values())]
masked = [i for i, q in enumerate(qs) if not all(0 < w < 1000 for w in q.values())]
print("%d plain / %d masked queries of %d" % (len(plain), len(masked), len(qs)))
import statistics
for tag, sel in (("plain", plain[:100]), ("masked", masked[:100])):
    ts = []
    for i in sel:
        t1 = time.perf_counter(); eng.score_topk(qs[i:i + 1], qv[i:i + 1], TOPK); ts.append((time.perf_counter() - t1) * 1e6)
    print("python path, %s queries: median %.1f us, mean %.1f us, p90 %.1f us" % (tag, statistics.median(ts), statistics.mean(ts), sorted(ts)[int(0.9 * len(ts))]))
q = qs[plain[0]]; qt = np.asarray(list(q.keys()), np.int32); qw = np.asarray(list(q.values()), np.float64); qp = np.asarray([0, len(qt)], np.int32)
ids = np.empty((1, TOPK), np.int32); vals = np.empty((1, TOPK), np.float64); v = np.ascontiguousarray(qv[plain[0]:plain[0] + 1])
lib = _lib.load(); fn = lib.hipts_search
args = (bm._h, idx._h, _lib.ptr(qt), _lib.ptr(qw), _lib.ptr(qp), _lib.ptr(v), 1, ctypes.c_double(0.5), ctypes.c_double(0.5), TOPK, _lib.ptr(ids), _lib.ptr(vals), None, None)
for _ in range(10): fn(*args)
t0 = time.perf_counter()
for _ in range(500): fn(*args)
print("C ABI call only, null stream: %.1f us / query" % ((time.perf_counter() - t0) / 500 * 1e6))
c1, c2 = ctypes.c_uint32(), ctypes.c_uint32()
lib.hiptsdbg_search1_last.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
lib.hiptsdbg_search1_last(bm._h, ctypes.byref(c1), ctypes.byref(c2))
print("last query: %d candidates, candidate path taken: %d" % (c1.value, c2.value))
for kk in (100, 1024):
    took = []
    for i in range(32):
        eng.score_topk(qs[i:i + 1], qv[i:i + 1], kk)
        lib.hiptsdbg_search1_last(bm._h, ctypes.byref(c1), ctypes.byref(c2)); took.append((c1.value, c2.value))
    print("k=%d candidates:" % kk, took[:12], "fast %d/32" % sum(t[1] for t in took))
