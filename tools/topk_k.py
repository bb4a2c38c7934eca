import ctypes, os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import _lib
NQ, D = 256, 100_000
rng = np.random.default_rng(0)
base = (rng.random((NQ, D)) * 0.9 + 0.05)
dev = torch.from_numpy(base).cuda()
for K in (1, 10, 100, 400, 1024):
    ids = torch.empty((NQ, K), dtype=torch.int32, device="cuda"); vals = torch.empty((NQ, K), dtype=torch.float64, device="cuda")
    for _ in range(3):
        _lib.call("hipts_topk", _lib.ptr(dev), NQ, ctypes.c_int64(D), K, _lib.ptr(ids), _lib.ptr(vals), _lib.DEVICE, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        _lib.call("hipts_topk", _lib.ptr(dev), NQ, ctypes.c_int64(D), K, _lib.ptr(ids), _lib.ptr(vals), _lib.DEVICE, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize()
    print("k=%4d: %.1f us per 256-query launch" % (K, 1e5 * (time.perf_counter() - t0)), flush=True)
