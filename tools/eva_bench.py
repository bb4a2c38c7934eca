#!/usr/bin/env python3
"""Development aid: EVA02-L/14 @448 images/s (gpurun only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.tagger import EvaTagger
cfg = dict(synth.EVA02_L14_448)
w = synth.eva_weights(cfg, seed=0)
for B in (10, 32):
    m = EvaTagger(cfg, w, max_batch=B)
    imgs = torch.randint(0, 256, (B, 448, 448, 3), dtype=torch.uint8, device="cuda")
    probs = torch.empty((B, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    for _ in range(2): m.forward_u8(imgs, probs=probs, want="probs")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 12
    for _ in range(n): m.forward_u8(imgs, probs=probs, want="probs")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = m.flops_per_image()
    print("batch %d: %.2f ms  %.0f images/s  %.1f TFLOP/s (%.1f GFLOP/img)" % (B, dt * 1e3, B / dt, B * fl / dt / 1e12, fl / 1e9))
    m.close()
