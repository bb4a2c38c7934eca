#!/usr/bin/env python3
"""Development aid (gpurun only): how long the CPU part of Predictor.predict -- selection read-back, tag strings -- takes per batch of 64,
and how many tags a line carries with the synthetic checkpoint (the end-to-end pipeline's consumer thread does this per batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import synth
from hiptagsearch.tagger import Predictor, format_lines
pr = Predictor(device=0, max_batch=64)
pr.load_model()
imgs = synth.images_u8(64, 448, seed=3)
_, probs = pr.tagger_model.forward_u8(imgs, want="probs")
t0 = time.perf_counter()
for _ in range(5):
    counts, ids, _ = pr.selector.run(probs, 0.3, True, 0.3, True)
t1 = time.perf_counter()
for _ in range(5):
    lines = format_lines(pr.tag_names, counts, ids)
t2 = time.perf_counter()
print("tags per image: mean %.0f max %d; selector.run %.2f ms, format_lines %.2f ms per batch of 64; line length %d" % (
    counts.sum(axis=1).mean(), counts.sum(axis=1).max(), (t1 - t0) / 5 * 1e3, (t2 - t1) / 5 * 1e3, len(lines[0])))
