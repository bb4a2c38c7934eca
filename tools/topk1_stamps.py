#!/usr/bin/env python3
"""Development aid (gpurun only, -DHIPTS_X_TOPK_STAMPS=0 build): phases of the one-query path's last kernel (topk_kernel<candidates>)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth, _lib
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
from hiptagsearch.search import SearchEngine
D, V, K = 100_000, 10_000, 300
ptr, terms = synth.tag_corpus(D, V, seed=42)
bm = BM25Index(ptr, terms, V, 0)
idx = Similarity("bench", None, K, 0, capacity=D)
idx.add_matrix(synth.index_vectors(D, K, seed=46))
eng = SearchEngine(None, idx, {}, bm, [])
qs = [dict(q) for q in synth.queries(64, V, seed=43)]
qv = np.random.default_rng(5).standard_normal((64, K)).astype(np.float32)
for i in range(16): eng.score_topk(qs[i:i + 1], qv[i:i + 1], 100)
acc = np.zeros(4)
for i in range(16, 48):
    eng.score_topk(qs[i:i + 1], qv[i:i + 1], 100)
    st = (ctypes.c_ulonglong * 16)()
    _lib.check(_lib.load().hiptsdbg_topk_stamps(st))
    t = {j: st[j] for j in (12, 13, 6, 7, 14, 15)}
    acc += np.array([t[13] - t[12], t[6] - t[13], t[7] - t[6], t[15] - t[7]]) / 100.0
acc /= 32
print("topk<candidates>: gather candidates %.1f us, state clear .. rank start %.1f us, rank + store %.1f us, fill + publish %.1f us; sum %.1f us" % (tuple(acc) + (acc.sum(),)))
