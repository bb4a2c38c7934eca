#!/usr/bin/env python3
"""Development aid: the one-query path over index sizes (does the candidate path decide the ranking? latency)."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth, _lib
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
from hiptagsearch.search import SearchEngine
lib = _lib.load()
for D in (10_000, 20_000, 50_000, 100_000, 250_000):
    V, K = 10_000, 300
    ptr, terms = synth.tag_corpus(D, V, seed=42)
    rows = synth.index_vectors(D, K, seed=46)
    bm = BM25Index(ptr, terms, V, 0)
    idx = Similarity("bench", None, K, 0, capacity=D); idx.add_matrix(rows)
    eng = SearchEngine(None, idx, {}, bm, [])
    qs = [dict(q) for q in synth.queries(64, V, seed=43)]
    qv = np.random.default_rng(5).standard_normal((64, K)); qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    for k in (100, 1024):
        cands, fast, ts = [], 0, []
        for i in range(64):
            t0 = time.perf_counter(); eng.score_topk(qs[i:i + 1], qv[i:i + 1], k); ts.append((time.perf_counter() - t0) * 1e6)
            c1, c2 = ctypes.c_uint32(), ctypes.c_uint32()
            _lib.call("hiptsdbg_search1_last", bm._h, ctypes.byref(c1), ctypes.byref(c2))
            cands.append(c1.value); fast += c2.value
        ts = sorted(ts[8:])
        print("D %7d k %4d: candidate path %2d/64, candidates median %5d max %5d; python call median %.1f us p90 %.1f us" % (
            D, k, fast, sorted(cands)[32], max(cands), ts[len(ts) // 2], ts[int(0.9 * len(ts))]), flush=True)
    del eng, bm, idx
