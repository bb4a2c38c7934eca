"""CPU: oracle/resize.py (numpy restatement of Pillow's 8-bit resample) against Pillow itself and against the committed digests of
Pillow's outputs (tests/golden/g10_resize.json, made by tests/golden/make_golden_resize.py)."""
import hashlib
import json
import os

import numpy as np
import pytest


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g10_resize.json")))["cases"]


def _input(c):
    a = np.random.default_rng(c["seed"]).integers(0, 256, (c["H"], c["W"], 3), dtype=np.uint8)
    return (a // 64) * 64 if c["posterise"] else a


def test_oracle_resize_matches_the_golden_digests(golden_dir):
    from oracle import resize as orz
    for c in _cases(golden_dir):
        if c["H"] * c["W"] > 1600 * 1600:
            continue                                                  # the 2048^2 case is the GPU test's (seconds of numpy here)
        got = orz.resize_u8(_input(c), c["out"], c["out"], c["filter"])
        assert [int(v) for v in got.reshape(-1)[:12]] == c["first_bytes"], c
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], c


def test_oracle_resize_matches_pillow_live():
    from PIL import Image
    from oracle import resize as orz
    rng = np.random.default_rng(99)
    for (h, w, oh, ow, kind) in [(333, 517, 448, 448, 3), (517, 333, 384, 384, 2), (64, 64, 448, 448, 3), (900, 900, 448, 448, 3), (448, 448, 448, 448, 3),
                                 (500, 448, 448, 448, 3), (448, 500, 384, 384, 2)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC if kind == 3 else Image.BILINEAR))
        np.testing.assert_array_equal(orz.resize_u8(a, oh, ow, kind), want)
