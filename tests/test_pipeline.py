"""Host input pipeline: the multi-process decode pool and the pre-decoded shard format hand the
device path exactly the uint8 images the in-process `Predictor.gen_image_tensor` produces."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))


def _make_images(d):
    from PIL import Image
    rng = np.random.default_rng(7)
    paths = []
    specs = [("a.png", "RGB", (448, 448)), ("b.png", "RGBA", (300, 500)), ("c.jpg", "RGB", (640, 360)), ("d.png", "LA", (123, 77)),
             ("e.png", "P", (200, 200)), ("f.png", "L", (448, 448)), ("g.png", "RGB", (31, 17))]
    for name, mode, (w, h) in specs:
        ch = {"RGB": 3, "RGBA": 4, "LA": 2, "L": 1, "P": 1}[mode]
        arr = rng.integers(0, 256, (h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
        img = Image.fromarray(arr, mode="L" if mode == "P" else mode)
        if mode == "P":
            img = img.convert("P")
        p = os.path.join(d, name)
        img.save(p)
        paths.append(p)
    bad = os.path.join(d, "broken.png")
    with open(bad, "wb") as f:
        f.write(b"not an image")
    return paths, bad


def test_decode_pool_matches_in_process_decode(tmp_path):
    from hiptagsearch import pipeline
    from hiptagsearch.tagger import Predictor
    paths, bad = _make_images(str(tmp_path))
    pred = Predictor.__new__(Predictor)
    pred.cfg = {"image_size": 448}
    want = {p: Predictor.gen_image_tensor(pred, p) for p in paths}
    order = paths[:3] + [bad] + paths[3:]
    got_paths, got = [], []
    with pipeline.DecodePool(workers=2, size=448, batch=3) as pool:
        for kept, images in pool.batches(order):
            assert images.dtype == np.uint8 and images.shape[1:] == (448, 448, 3) and len(kept) == images.shape[0]
            got_paths += kept
            got += [im.copy() for im in images]
    assert got_paths == paths                                   # file order kept, the broken file dropped
    for p, im in zip(got_paths, got):
        np.testing.assert_array_equal(im, want[p])
    # CCIP variant: no padding, bilinear to 384 (gen_cfeatures.py:285-295 before the float normalisation)
    from PIL import Image
    a = pipeline.decode_image(paths[1], 384, pipeline.CCIP)
    img = Image.open(paths[1]); img.load()
    bg = Image.new("RGB", img.size, (255, 255, 255)); bg.paste(img, mask=img.split()[-1])
    np.testing.assert_array_equal(a, np.asarray(bg.resize((384, 384), resample=Image.BILINEAR)))


def test_shards_round_trip(tmp_path):
    from hiptagsearch import pipeline
    src = tmp_path / "src"
    src.mkdir()
    paths, bad = _make_images(str(src))
    out = str(tmp_path / "shards")
    n = pipeline.write_shards(paths[:4] + [bad] + paths[4:], out, size=64, workers=2, per_shard=3, batch=2)
    assert n == len(paths)
    assert sorted(os.listdir(out)) == ["shard-00000.npy", "shard-00000.txt", "shard-00001.npy", "shard-00001.txt", "shard-00002.npy", "shard-00002.txt"]
    got_paths, got = [], []
    for kept, images in pipeline.iter_shards(out, batch=2):
        got_paths += kept
        got += list(images)
    assert got_paths == paths
    for p, im in zip(got_paths, got):
        np.testing.assert_array_equal(im, pipeline.decode_image(p, 64))
