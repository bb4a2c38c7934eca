#!/usr/bin/env python3
"""Development aid (gpurun only): ViT-B/16 @448 batch 64 with the images handed over as HOST buffers (pageable numpy, pinned torch tensor) against
device-resident input -- the PCIe-inclusive rate of the boundary."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger
cfg = dict(synth.VIT_B16_448)
m = ViTTagger(cfg, synth.vit_weights(cfg, seed=0), max_batch=64)
B = 64
host = np.random.default_rng(0).integers(0, 256, (B, 448, 448, 3), dtype=np.uint8)
pinned = torch.from_numpy(host).pin_memory()
dev = torch.from_numpy(host).cuda()
probs = torch.empty((B, cfg["num_classes"]), dtype=torch.float32, device="cuda")
for name, x in (("device-resident u8", dev), ("pinned host u8", pinned), ("pageable host u8 (numpy)", host)):
    for _ in range(3): m.forward_u8(x, probs=probs, want="probs")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 12
    for _ in range(n): m.forward_u8(x, probs=probs, want="probs")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-28s %.2f ms per batch of 64  %.0f images/s" % (name, dt * 1e3, B / dt), flush=True)
