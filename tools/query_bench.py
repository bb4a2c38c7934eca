#!/usr/bin/env python3
"""Development aid: time the query path alone (gpurun / rocprofv3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
import bench
t0 = time.perf_counter()
r = bench.query_section(0)
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}, "total %.1fs" % (time.perf_counter() - t0))
