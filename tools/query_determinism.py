#!/usr/bin/env python3
"""Development aid (gpurun only): the same 256-query batch through the fused search 60 times, the same index product 20 times -- every
result bit-identical (a race in the LDS candidate lists / packing / the wide product's staging shows up as run-to-run differences
before it shows up as a wrong answer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
from hiptagsearch.search import SearchEngine
D, V, K = 100_000, 10_000, 300
ptr, terms = synth.tag_corpus(D, V, seed=42)
bm = BM25Index(ptr, terms, V, 0)
idx = Similarity("bench", None, K, 0, capacity=D)
idx.add_matrix(synth.index_vectors(D, K, seed=46))
eng = SearchEngine(None, idx, {}, bm, [])
qs = [dict(q) for q in synth.queries(256, V, seed=43)]
qv = np.random.default_rng(5).standard_normal((256, K)).astype(np.float32)
bad = 0
for k in (100, 1024, 7):
    ref = eng.score_topk(qs, qv, k)
    for it in range(20):
        cur = eng.score_topk(qs, qv, k)
        if not (np.array_equal(cur[0], ref[0]) and cur[1].tobytes() == ref[1].tobytes()):
            bad += 1
            print("k=%d run %d differs: %d ids, %d values" % (k, it, int((cur[0] != ref[0]).sum()), int((cur[1] != ref[1]).sum())), flush=True)
ref = idx.query(qv)
for it in range(20):
    cur = idx.query(qv)
    if cur.tobytes() != ref.tobytes():
        bad += 1
        print("index product run %d differs in %d scores" % (it, int((cur != ref).sum())), flush=True)
one = [eng.score_topk(qs[i:i + 1], qv[i:i + 1], 100) for i in range(64)]
for it in range(5):
    for i in range(64):
        cur = eng.score_topk(qs[i:i + 1], qv[i:i + 1], 100)
        if not (np.array_equal(cur[0], one[i][0]) and cur[1].tobytes() == one[i][1].tobytes()):
            bad += 1
            print("one-query path, query %d run %d differs" % (i, it), flush=True)
print("DETERMINISM", "OK" if bad == 0 else "FAILED (%d)" % bad)
sys.exit(1 if bad else 0)
