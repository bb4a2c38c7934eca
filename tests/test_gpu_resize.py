"""GPU: hipts_resize_u8 (csrc/resize.hip) is Pillow's resample bit for bit -- against the committed digests of Pillow's outputs, against
Pillow live, and through the product's input path (Predictor.gen_image_tensor with gpu_resize)."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _resize(a, oh, ow, filt, device_src=False):
    import torch
    from hiptagsearch import _lib
    out = torch.empty((oh, ow, 3), dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(a).cuda() if device_src else np.ascontiguousarray(a)
    _lib.call("hipts_resize_u8", _lib.ptr(src), _lib.memspace_of(src), a.shape[0], a.shape[1], _lib.ptr(out), oh, ow, filt, 0, _lib.current_stream_ptr())
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_resize_matches_the_golden_digests(golden_dir):
    for c in json.load(open(os.path.join(golden_dir, "g10_resize.json")))["cases"]:
        a = np.random.default_rng(c["seed"]).integers(0, 256, (c["H"], c["W"], 3), dtype=np.uint8)
        if c["posterise"]:
            a = (a // 64) * 64
        got = _resize(a, c["out"], c["out"], c["filter"], device_src=bool(c["seed"] & 1))
        assert [int(v) for v in got.reshape(-1)[:12]] == c["first_bytes"], c
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], c


def test_resize_matches_pillow_live_including_one_axis_cases():
    from PIL import Image
    rng = np.random.default_rng(5)
    for (h, w, oh, ow, kind) in [(333, 517, 448, 448, 3), (517, 333, 384, 384, 2), (448, 448, 448, 448, 3), (500, 448, 448, 448, 3), (448, 500, 384, 500, 2),
                                 (1200, 1600, 448, 448, 3), (31, 29, 448, 448, 3), (3000, 3000, 448, 448, 3)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BICUBIC if kind == 3 else Image.BILINEAR))
        np.testing.assert_array_equal(_resize(a, oh, ow, kind), want)


def test_predictor_gpu_resize_gives_the_host_resize_image(tmp_path):
    """Predictor.gen_image_tensor(..., gpu_resize=True): decode + prepare_image on the host, the Resize(bicubic) on the device --
    the same uint8 image as the host path, hence the same tags."""
    from PIL import Image
    from hiptagsearch import synth
    from hiptagsearch.tagger import Predictor
    rng = np.random.default_rng(8)
    paths = []
    for i, (h, w) in enumerate([(300, 500), (700, 640), (448, 448), (90, 40)]):
        p = str(tmp_path / ("im%d.png" % i))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    pr = Predictor(max_batch=4)
    pr.load_model(cfg=dict(synth.VIT_TINY))
    for p in paths:
        host = pr.gen_image_tensor(p)
        dev = pr.gen_image_tensor(p, gpu_resize=True)
        np.testing.assert_array_equal(np.asarray(dev.cpu()), host)
    a = pr.predict([pr.gen_image_tensor(p) for p in paths], 0.3, True, 0.3, True)
    b = pr.predict([np.asarray(pr.gen_image_tensor(p, gpu_resize=True).cpu()) for p in paths], 0.3, True, 0.3, True)
    assert a == b


def test_tagging_cli_gpu_resize_writes_the_same_file(tmp_path):
    import subprocess, sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "anime-illust-image-searcher_amd", "tagging.py")
    rng = np.random.default_rng(2)
    os.makedirs(tmp_path / "imgs")
    for i in range(13):
        Image.fromarray(rng.integers(0, 256, (120 + 17 * i, 200 - 9 * i, 3), dtype=np.uint8)).save(str(tmp_path / "imgs" / ("p%02d.png" % i)))
    outs = []
    for extra in ([], ["--gpu-resize"], ["--gpu-resize", "--workers", "2"]):
        if os.path.exists(tmp_path / "tags-wd-tagger.txt"):
            os.remove(tmp_path / "tags-wd-tagger.txt")
        r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--model", "vit-tiny", "--batch", "8"] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0].splitlines()) == 13


@pytest.mark.parametrize("mode", ["tagger", "ccip"])
def test_decode_pool_with_device_resize_yields_the_host_images(tmp_path, mode):
    """pipeline.DecodePool(device_resize=True): the workers only decode and composite, the consumer pads and resizes on the device --
    the same uint8 images as decode_image on the host, for RGB / RGBA / L / palette inputs, tall and wide and already-square ones, an
    image larger than a ring slot (resized by the worker) and a file that does not decode (dropped)."""
    from PIL import Image
    from hiptagsearch import pipeline
    rng = np.random.default_rng(11)
    size = 448 if mode == "tagger" else 384
    paths = []
    def save(name, img):
        p = str(tmp_path / name)
        img.save(p)
        paths.append(p)
    save("a.png", Image.fromarray(rng.integers(0, 256, (300, 500, 3), dtype=np.uint8)))
    save("b.png", Image.fromarray(rng.integers(0, 256, (520, 333, 4), dtype=np.uint8), "RGBA"))
    save("c.png", Image.fromarray(rng.integers(0, 256, (size, size), dtype=np.uint8), "L"))
    save("d.jpg", Image.fromarray(rng.integers(0, 256, (600, 800, 3), dtype=np.uint8)))
    open(tmp_path / "e.png", "wb").write(b"not an image")
    paths.append(str(tmp_path / "e.png"))
    save("f.png", Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).convert("P"))
    save("g.png", Image.fromarray(rng.integers(0, 256, (900, 700, 3), dtype=np.uint8)))        # 630 000 pixels > max_pixels below
    save("h.png", Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)))
    want = {p: pipeline.decode_image(p, size, mode) for p in paths}
    got = {}
    with pipeline.DecodePool(workers=2, size=size, batch=3, mode=mode, device_resize=True, max_pixels=500_000) as pool:
        for kept, images in pool.batches(paths):
            assert images.is_cuda and tuple(images.shape[1:]) == (size, size, 3) and len(kept) == images.shape[0]
            for p, img in zip(kept, images.cpu().numpy()):
                got[p] = img
    assert sorted(got) == sorted(p for p in paths if want[p] is not None) and len(got) == len(paths) - 1
    for p in got:
        np.testing.assert_array_equal(got[p], want[p], err_msg=p)
