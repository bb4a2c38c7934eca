#!/usr/bin/env python3
"""Development aid (gpurun only, -DHIPTS_X_TOPK_STAMPS build): where one workgroup of the batched top-k spends its time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import _lib
lib = _lib.load()
NQ, D = 256, 100_000
base = np.random.default_rng(0).random((NQ, D)) * 0.9 + 0.05
dev = torch.from_numpy(base).cuda()
names = ["start", "sample loads", "hist add", "threshold scan", "collect pass", "pack", "(exact path)", "rank + store"]
for K in (1, 100, 1024):
    ids = torch.empty((NQ, K), dtype=torch.int32, device="cuda"); vals = torch.empty((NQ, K), dtype=torch.float64, device="cuda")
    for _ in range(5):
        _lib.call("hipts_topk", _lib.ptr(dev), NQ, ctypes.c_int64(D), K, _lib.ptr(ids), _lib.ptr(vals), _lib.DEVICE, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    _lib.check(lib.hiptsdbg_topk_stamps(st))
    t = [st[i] for i in range(8)]
    print("k=%d: " % K + ", ".join("%s %.1f us" % (names[i], (t[i] - t[i - 1]) / 100.0) for i in range(1, 8)) + "; total %.1f us" % ((t[7] - t[0]) / 100.0), flush=True)
