#!/usr/bin/env python3
"""Development aid (gpurun only): host enqueue time of a CCIP / ViT / EVA02 forward against its device time (is the forward launch-bound?)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.cfeatures import CCIPEncoder
w = synth.ccip_weights(dict(synth.CCIP_B36_384), seed=46)
for B in (1, 4, 20, 64):
    enc = CCIPEncoder(dict(synth.CCIP_B36_384), w, max_batch=B)
    imgs = torch.randint(0, 256, (B, 384, 384, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, 768), dtype=torch.float32, device="cuda")
    for _ in range(3): enc.forward_u8(imgs, out=out)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter(); host = 0.0
    for _ in range(n):
        h0 = time.perf_counter(); enc.forward_u8(imgs, out=out); host += time.perf_counter() - h0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("ccip batch %3d: %.2f ms per forward, host enqueue %.2f ms" % (B, dt * 1e3, host / n * 1e3), flush=True)
    del enc
