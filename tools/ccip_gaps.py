#!/usr/bin/env python3
"""Development aid (gpurun only, under rocprofv3 --kernel-trace): CCIP forwards at one batch size; the trace then shows how much of a forward's wall
time is inside kernels (tools/gpurun/r4_ccip_gaps.sh sums it)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.cfeatures import CCIPEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = synth.ccip_weights(dict(synth.CCIP_B36_384), seed=46)
enc = CCIPEncoder(dict(synth.CCIP_B36_384, operand_f16=1), w, max_batch=B)
imgs = torch.randint(0, 256, (B, 384, 384, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((B, 768), dtype=torch.float32, device="cuda")
for _ in range(3): enc.forward_u8(imgs, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): enc.forward_u8(imgs, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("batch %d: %.3f ms per forward (wall, %d forwards back to back)" % (B, dt * 1e3, n))
