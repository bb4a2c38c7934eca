#!/usr/bin/env python3
"""Development aid (gpurun only): hipts_topk (batched, 256 queries x 100 k scores, k = 100) by kind of score row -- all finite, or all but
N finite scores -inf (what a required term leaves) -- to see which rows set the launch time (one workgroup per query, one round)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import _lib
NQ, D, K = 256, 100_000, 100
rng = np.random.default_rng(0)
ids = torch.empty((NQ, K), dtype=torch.int32, device="cuda"); vals = torch.empty((NQ, K), dtype=torch.float64, device="cuda")
def run(name, rows):
    dev = torch.from_numpy(rows).cuda()
    for _ in range(3):
        _lib.call("hipts_topk", _lib.ptr(dev), NQ, ctypes.c_int64(D), K, _lib.ptr(ids), _lib.ptr(vals), _lib.DEVICE, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        _lib.call("hipts_topk", _lib.ptr(dev), NQ, ctypes.c_int64(D), K, _lib.ptr(ids), _lib.ptr(vals), _lib.DEVICE, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize()
    us = 1e5 * (time.perf_counter() - t0)
    got = ids[0].cpu().numpy(); want = np.lexsort((np.arange(D), -rows[0]))[:K]
    print("%-34s %7.1f us per 256-query launch   correct %s" % (name, us, bool(np.array_equal(got, want))), flush=True)
base = (rng.random((NQ, D)) * 0.9 + 0.05)
run("all finite", base.copy())
for nf in (20, 99, 101, 300, 3000, 30000):
    r = np.full((NQ, D), -np.inf)
    for q in range(NQ):
        idx = rng.choice(D, nf, replace=False)
        r[q, idx] = base[q, idx]
    run("%d finite, rest -inf" % nf, r)
r = base.copy(); r[:, ::2] = -np.inf
run("half -inf (an excluded term)", r)
r = np.round(base, 2)
run("two-decimal ties", r)
mix = base.copy()
for q in range(0, NQ, 7):
    idx = rng.choice(D, 50, replace=False); t = np.full(D, -np.inf); t[idx] = base[q, idx]; mix[q] = t
run("mostly finite, every 7th 50-finite", mix)
