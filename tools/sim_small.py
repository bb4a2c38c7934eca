import ctypes, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/anime-illust-image-searcher_amd")
import numpy as np, torch
from hiptagsearch.index import Similarity
rng = np.random.default_rng(0)
for D in (1000, 10000, 100000):
    rows = rng.standard_normal((D, 300)).astype(np.float32)
    idx = Similarity("t", None, 300); idx.add_matrix(rows)
    q = rng.standard_normal((256, 300)).astype(np.float32)
    for _ in range(3): idx.query(q)
    t0 = time.perf_counter()
    for _ in range(20): idx.query(q)
    print("D=%6d: %.1f us per 256-query index product (incl. D2H of the scores)" % (D, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
