#!/usr/bin/env python3
"""Development aid (gpurun only, -DHIPTS_X_TOPK_STAMPS=<workgroup> build): phases of one workgroup (= one query) of bm25_postings_kernel in a
256-query batch of the bench corpus."""
import ctypes, os, sys
os.environ["HIPTS_DBG_BM25_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth, _lib
from hiptagsearch.bm25 import BM25Index
from hiptagsearch.index import Similarity
from hiptagsearch.search import SearchEngine
D, V, K, TOPK = 100_000, 10_000, 300, 100
ptr, terms = synth.tag_corpus(D, V, seed=42)
bm = BM25Index(ptr, terms, V, 0)
idx = Similarity("bench", None, K, 0, capacity=D)
idx.add_matrix(synth.index_vectors(D, K, seed=46))
eng = SearchEngine(None, idx, {}, bm, [])
qs = [dict(q) for q in synth.queries(256, V, seed=43)]
qv = np.random.default_rng(5).standard_normal((256, K)).astype(np.float32)
for _ in range(3): eng.score_topk(qs, qv, TOPK)
torch.cuda.synchronize()
st = (ctypes.c_ulonglong * 16)()
_lib.check(_lib.load().hiptsdbg_topk_stamps(st))
t = [st[8 + i] for i in range(4)]
wg = int(os.environ.get("WG", "0"))
df = np.bincount(terms, minlength=V)
q = qs[wg]
print("query %d: terms %s, sum df %d; clear %.1f us, postings %.1f us, mask + max %.1f us, total %.1f us" % (
    wg, {k: round(v, 1) for k, v in q.items()}, sum(int(df[k]) for k in q if 0 <= k < V), (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0, (t[3] - t[0]) / 100.0))
