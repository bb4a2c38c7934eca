#!/usr/bin/env python3
"""Development aid: CCIP encoder images/s at the reference batch (20) and at 64 (gpurun only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.cfeatures import CCIPEncoder
w = synth.ccip_weights(dict(synth.CCIP_B36_384), seed=46)
modes = [int(a) for a in sys.argv[1:]] or [1]      # 0 bf16, 1 half (2, the e4m3 mode, was withdrawn in round 4)
for mode, B in [(m, b) for m in modes for b in (20, 64)]:
    cfg = dict(synth.CCIP_B36_384, operand_f16=mode)
    enc = CCIPEncoder(cfg, w, max_batch=B)
    imgs = torch.randint(0, 256, (B, 384, 384, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, 768), dtype=torch.float32, device="cuda")
    for _ in range(2): enc.forward_u8(imgs, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): enc.forward_u8(imgs, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = enc.flops_per_image()
    print("operands %s batch %d: %.2f ms  %.0f images/s  %.1f TFLOP/s (%.2f GFLOP/img)" % (("bf16", "half")[mode], B, dt * 1e3, B / dt, B * fl / dt / 1e12, fl / 1e9))
    del enc
