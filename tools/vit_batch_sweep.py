#!/usr/bin/env python3
"""Development aid: ViT-B/16 @448 images/s over batch sizes (gpurun only).  HIPTS_VIT_STREAMS is read once per process:
run it once per stream count.   usage: vit_batch_sweep.py B [B ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import torch
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger
cfg = dict(synth.VIT_B16_448)
w = synth.vit_weights(cfg, seed=0)
for B in [int(x) for x in sys.argv[1:]]:
    m = ViTTagger(cfg, w, max_batch=B)
    imgs = torch.randint(0, 256, (B, 448, 448, 3), dtype=torch.uint8, device="cuda")
    probs = torch.empty((B, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    for _ in range(4): m.forward_u8(imgs, probs=probs, want="probs")
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        n = 12
        for _ in range(n): m.forward_u8(imgs, probs=probs, want="probs")
        torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / n)
    dt = sorted(best)[1]
    print("streams %s batch %3d: %.2f ms  %.0f images/s  (%.3f ms/image)" % (os.environ.get("HIPTS_VIT_STREAMS", "2"), B, dt * 1e3, B / dt, dt * 1e3 / B), flush=True)
    m.close()
