#!/usr/bin/env python3
"""Generates tests/golden/g10_resize.json: sha256 digests of Pillow's own `Image.resize` outputs (the call behind the timm eval
transform of tagging.py:241 and gen_cfeatures.py:101) on seeded inputs -- the fixture that pins oracle/resize.py and, through it and
directly, the device kernel hipts_resize_u8.  Run in the build container (Pillow 12.2 is installed); commits data, not source."""
import hashlib
import json
import os

import numpy as np
from PIL import Image

CASES = [  # (seed, H, W, out, PIL filter enumerator, posterise)
    (1, 600, 800, 448, 3, False), (2, 800, 600, 448, 3, False), (3, 1024, 1024, 448, 3, True), (4, 100, 100, 448, 3, False),
    (5, 449, 449, 448, 3, False), (6, 1500, 1500, 448, 3, False), (7, 713, 713, 384, 2, False), (8, 50, 70, 384, 2, False),
    (9, 1000, 1000, 384, 2, True), (10, 448, 300, 448, 3, False), (11, 2048, 2048, 448, 3, False), (12, 384, 384, 384, 2, False),
]


def case_input(seed, H, W, posterise):
    a = np.random.default_rng(seed).integers(0, 256, (H, W, 3), dtype=np.uint8)
    return (a // 64) * 64 if posterise else a


def main():
    out = {"pillow_version": Image.__version__ if hasattr(Image, "__version__") else "", "cases": []}
    import PIL
    out["pillow_version"] = PIL.__version__
    for seed, H, W, size, filt, post in CASES:
        a = case_input(seed, H, W, post)
        r = np.asarray(Image.fromarray(a).resize((size, size), Image.BICUBIC if filt == 3 else Image.BILINEAR))
        out["cases"].append({"seed": seed, "H": H, "W": W, "out": size, "filter": filt, "posterise": post,
                             "sha256": hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest(), "first_bytes": [int(v) for v in r.reshape(-1)[:12]]})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g10_resize.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
