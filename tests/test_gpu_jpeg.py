"""GPU parity of the hybrid JPEG decode (csrc/jpeg_host.c + csrc/jpeg.hip) through the C ABI: the bytes PIL.Image.open gives the
reference (tagging.py:234-252, gen_cfeatures.py:285-295) -- libjpeg-turbo with its defaults -- reproduced exactly, alone, in front of the
pad + resize, and inside the multi-process decode pipeline."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_oracle_jpeg import cases, jpeg_bytes, synth_image  # noqa: E402


def _slot_of(_lib, data, stride=None):
    w, h = Image.open(io.BytesIO(data)).size
    nb = stride or int(_lib.load().hipts_jpeg_slot_bytes(w, h))
    slot = np.zeros(nb, dtype=np.uint8)
    src = np.frombuffer(data, dtype=np.uint8)
    st = _lib.load().hipts_jpeg_entropy_decode(src.ctypes.data, len(data), slot.ctypes.data, nb)
    return st, slot, (h, w)


def test_device_decode_equals_pillow_and_oracle():
    from hiptagsearch import _lib
    from oracle import jpeg as oj
    import torch
    rng = np.random.default_rng(0)
    for (h, w), grey, kw in cases():
        data = jpeg_bytes(synth_image(rng, h, w, grey), **kw)
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        st, slot, _ = _slot_of(_lib, data)
        assert st == 0
        got = np.empty((h, w, 3), dtype=np.uint8)
        _lib.call("hipts_jpeg_decode_rgb", slot.ctypes.data, slot.nbytes, got.ctypes.data, _lib.HOST, got.nbytes, 0, None)
        assert np.array_equal(got, want), ((h, w), grey, kw, int(np.abs(got.astype(int) - want).max()))
        assert np.array_equal(got, oj.decode_slot(slot))
    # device-resident output
    dev = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    _lib.call("hipts_jpeg_decode_rgb", slot.ctypes.data, slot.nbytes, _lib.ptr(dev), _lib.DEVICE, dev.numel(), 0, None)
    assert np.array_equal(dev.cpu().numpy(), want)
    # a slot that is not one, an output that is too small
    bad = slot.copy()
    bad[:4] = 0
    with pytest.raises(_lib.HipTagSearchError):
        _lib.call("hipts_jpeg_decode_rgb", bad.ctypes.data, bad.nbytes, got.ctypes.data, _lib.HOST, got.nbytes, 0, None)
    with pytest.raises(_lib.HipTagSearchError):
        _lib.call("hipts_jpeg_decode_rgb", slot.ctypes.data, slot.nbytes, got.ctypes.data, _lib.HOST, got.nbytes - 1, 0, None)


@pytest.mark.parametrize("mode", ["tagger", "ccip"])
def test_batch_of_coefficient_and_decoded_slots(mode):
    """hipts_jpeg_batch_u8 = decode on the device + hipts_resize_batch_u8: against the latter fed with Pillow's decode of the same files
    (itself pinned to Pillow's resize in test_gpu_resize.py), coefficient slots and already-decoded slots mixed."""
    from hiptagsearch import _lib
    import torch
    rng = np.random.default_rng(4)
    S = 448 if mode == "tagger" else 384
    files = []
    for (h, w), sub, q in [((768, 1024), 2, 90), ((500, 333), 1, 80), ((97, 211), 0, 95), ((1200, 900), 2, 85), ((64, 64), 2, 50)]:
        files.append(jpeg_bytes(synth_image(rng, h, w), quality=q, subsampling=sub))
    files.append(jpeg_bytes(synth_image(rng, 300, 400).convert("CMYK"), quality=85))       # refused by the fast path: decoded slot
    files.append(jpeg_bytes(synth_image(rng, 240, 320, grey=True), quality=85))
    files.append(jpeg_bytes(synth_image(rng, 333, 222), quality=80, subsampling=2, progressive=True))
    stride = 1200 * 900 * 4
    slots = np.zeros((len(files), stride), dtype=np.uint8)
    ref = np.zeros((len(files), stride), dtype=np.uint8)
    kinds, hw = [], []
    for i, data in enumerate(files):
        img = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        ref[i, :img.size] = img.reshape(-1)
        st, slot, (h, w) = _slot_of(_lib, data, stride)
        if st == 0:
            slots[i] = slot
            kinds.append(1)
        else:
            assert st == 1
            slots[i, :img.size] = img.reshape(-1)
            kinds.append(0)
        hw.append((h, w))
    assert kinds == [1, 1, 1, 1, 1, 0, 1, 1]
    kinds = np.asarray(kinds, dtype=np.int32)
    hw = np.ascontiguousarray(np.asarray(hw, dtype=np.int32))
    pad, filt = (1, 3) if mode == "tagger" else (0, 2)
    got = torch.empty((len(files), S, S, 3), dtype=torch.uint8, device="cuda")
    want = torch.empty_like(got)
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("hipts_jpeg_batch_u8", slots.ctypes.data, stride, _lib.ptr(kinds), _lib.ptr(hw), len(files), pad, _lib.ptr(got), S, filt, 0, s)
    _lib.call("hipts_resize_batch_u8", ref.ctypes.data, _lib.HOST, stride, _lib.ptr(hw), len(files), pad, _lib.ptr(want), S, filt, 0, s)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # twice in a row on two streams (the staging buffers are shared: the second call is ordered behind the first)
    side = torch.cuda.Stream()
    got2 = torch.empty_like(got)
    _lib.call("hipts_jpeg_batch_u8", slots.ctypes.data, stride, _lib.ptr(kinds), _lib.ptr(hw), len(files), pad, _lib.ptr(got2), S, filt, 0, side.cuda_stream)
    _lib.call("hipts_jpeg_batch_u8", slots.ctypes.data, stride, _lib.ptr(kinds), _lib.ptr(hw), len(files), pad, _lib.ptr(got), S, filt, 0, s)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got2, want)
    # a header that disagrees with the caller is refused
    hw2 = hw.copy()
    hw2[0, 0] += 1
    with pytest.raises(_lib.HipTagSearchError):
        _lib.call("hipts_jpeg_batch_u8", slots.ctypes.data, stride, _lib.ptr(kinds), _lib.ptr(hw2), len(files), pad, _lib.ptr(got), S, filt, 0, s)


def test_decode_pool_with_device_jpeg_equals_pillow_workers(tmp_path):
    """The multi-process pipeline with the hybrid decode against the same pipeline decoding with Pillow: same kept files, same uint8
    model inputs -- JPEGs of three samplings, a PNG with alpha, a progressive JPEG, a truncated JPEG (dropped by both), a big JPEG
    (beyond max_pixels: resized by the worker in both)."""
    from hiptagsearch import pipeline
    import torch
    rng = np.random.default_rng(9)
    paths = []
    for i in range(40):
        h, w = int(rng.integers(40, 700)), int(rng.integers(40, 900))
        p = str(tmp_path / ("img%03d.jpg" % i))
        synth_image(rng, h, w, grey=(i % 13 == 5)).save(p, quality=int(rng.integers(40, 98)), **({} if i % 13 == 5 else {"subsampling": i % 3}))
        paths.append(p)
    p = str(tmp_path / "alpha.png")
    Image.fromarray(rng.integers(0, 256, (90, 120, 4), dtype=np.uint8), "RGBA").save(p)
    paths.insert(7, p)
    p = str(tmp_path / "progressive.jpg")
    synth_image(rng, 300, 200).save(p, quality=85, progressive=True)
    paths.insert(20, p)
    p = str(tmp_path / "cmyk.jpg")
    synth_image(rng, 150, 210).convert("CMYK").save(p, quality=85)
    paths.insert(25, p)
    p = str(tmp_path / "truncated.jpg")
    data = jpeg_bytes(synth_image(rng, 300, 200), quality=85)
    open(p, "wb").write(data[:len(data) // 2])
    paths.insert(30, p)
    p = str(tmp_path / "big.jpg")
    synth_image(rng, 700, 900).save(p, quality=85)
    paths.append(p)
    outs = {}
    for jpeg in (False, True):
        kept_all, tens = [], []
        with pipeline.DecodePool(workers=4, size=448, batch=16, mode=pipeline.TAGGER, device_resize=True, max_pixels=600 * 800, device_jpeg=jpeg) as pool:
            for kept, images in pool.batches(paths):
                kept_all += kept
                tens.append(images.clone())
        torch.cuda.synchronize()
        outs[jpeg] = (kept_all, torch.cat(tens).cpu().numpy())
    assert outs[True][0] == outs[False][0] and len(outs[True][0]) == len(paths) - 1
    assert np.array_equal(outs[True][1], outs[False][1])
