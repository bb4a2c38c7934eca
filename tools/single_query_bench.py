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
