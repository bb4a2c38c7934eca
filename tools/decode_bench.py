#!/usr/bin/env python3
"""Development aid: host decode throughput, the reference's 8 threads vs the multi-process DecodePool (no GPU needed)."""
import concurrent.futures, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from PIL import Image
from hiptagsearch import pipeline

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    paths = []
    for i in range(n):      # smooth-ish 1024 x 768 JPEGs (a typical illustration is larger than the model input)
        img = Image.fromarray(np.roll(base, i, axis=0)).resize((1024, 768), Image.BICUBIC)
        p = os.path.join(d, "%05d.jpg" % i)
        img.save(p, quality=90)
        paths.append(p)
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        out = list(ex.map(lambda p: pipeline.decode_image(p, 448), paths))
    dt = time.perf_counter() - t0
    print("8 threads (reference layout): %.0f images/s" % (n / dt))
    for w in (8, os.cpu_count() or 8):
        with pipeline.DecodePool(workers=w, size=448, batch=64) as pool:
            list(pool.batches(paths[:64]))           # worker start-up
            t0 = time.perf_counter()
            m = sum(len(k) for k, _ in pool.batches(paths))
            dt = time.perf_counter() - t0
        print("DecodePool %d processes: %.0f images/s" % (w, m / dt))
