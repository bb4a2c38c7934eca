"""CPU: the host half of the hybrid JPEG decode (csrc/jpeg_host.c -> libhipts_jpeg_host.so: markers + Huffman entropy decoding) followed by
the oracle's restatement of libjpeg's inverse DCT / fancy upsampling / colour conversion (oracle/jpeg.py) gives the bytes of
PIL.Image.open(file).convert('RGB') -- the decode the reference performs (tagging.py:234-252) -- on files Pillow writes here: three chroma
samplings, odd sizes, qualities 30..100 (quality 100: all-ones quantisation tables, long codes), optimised Huffman tables, restart
markers, greyscale.  Files the fast path does not take (progressive, CMYK, tiny, truncated) are refused with a status, never decoded
wrongly.  This pins oracle/jpeg.py (the checker of the GPU kernels, tests/test_gpu_jpeg.py) against libjpeg-turbo itself."""
import ctypes
import io
import os

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "anime-illust-image-searcher_amd", "libhipts_jpeg_host.so")


def host_lib():
    lib = ctypes.CDLL(HOST_LIB)
    lib.hipts_jpeg_entropy_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.hipts_jpeg_slot_bytes.restype = ctypes.c_int64
    return lib


def synth_image(rng, h, w, grey=False, noise=12):
    small = rng.integers(0, 256, (max(2, h // 32), max(2, w // 32), 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC), dtype=np.int16) + rng.integers(-noise, noise + 1, (h, w, 3), dtype=np.int16)
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    return im.convert("L") if grey else im


def jpeg_bytes(im, **kw):
    buf = io.BytesIO()
    im.save(buf, "JPEG", **kw)
    return buf.getvalue()


def entropy_decode(lib, data, slot_bytes=None):
    nb = slot_bytes or lib.hipts_jpeg_slot_bytes(*Image.open(io.BytesIO(data)).size)
    slot = np.zeros(nb, dtype=np.uint8)
    arr = np.frombuffer(data, dtype=np.uint8)
    return lib.hipts_jpeg_entropy_decode(arr.ctypes.data, len(data), slot.ctypes.data, nb), slot


def cases():
    out = []
    for (h, w) in [(768, 1024), (101, 77), (16, 16), (33, 250), (480, 641)]:
        for sub in (0, 1, 2):
            for q in (30, 90, 100):
                out.append(((h, w), False, dict(quality=q, subsampling=sub)))
    out.append(((300, 200), True, dict(quality=85)))
    out.append(((301, 203), False, dict(quality=85, subsampling=2, optimize=True)))
    out.append(((301, 203), False, dict(quality=85, subsampling=2, restart_marker_blocks=7)))
    out.append(((301, 203), False, dict(quality=75, subsampling=1, restart_marker_rows=1)))
    out.append(((64, 48), False, dict(quality=95, subsampling=0, optimize=True, restart_marker_blocks=1)))
    return out


def test_entropy_decode_then_oracle_equals_pillow():
    from oracle import jpeg as oj
    lib = host_lib()
    rng = np.random.default_rng(0)
    for (h, w), grey, kw in cases():
        data = jpeg_bytes(synth_image(rng, h, w, grey), **kw)
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        st, slot = entropy_decode(lib, data)
        assert st == 0, ((h, w), grey, kw, st)
        got = oj.decode_slot(slot)
        assert got.shape == want.shape
        assert np.array_equal(got, want), ((h, w), grey, kw, int(np.abs(got.astype(int) - want).max()))


def test_flat_and_saturated_content():
    """all-zero AC blocks (both zero-coefficient short cuts of jidctint.c are value-identical to the full passes), saturated colours
    (the range limit), one-colour chroma"""
    from oracle import jpeg as oj
    lib = host_lib()
    imgs = [Image.new("RGB", (64, 40), (255, 255, 255)), Image.new("RGB", (64, 40), (0, 0, 0)), Image.new("RGB", (50, 70), (255, 0, 0)),
            Image.new("RGB", (50, 70), (0, 0, 255))]
    a = np.zeros((96, 96, 3), np.uint8)
    a[::2, ::2] = 255
    a[1::2, 1::2, 1] = 255
    imgs.append(Image.fromarray(a))
    for im in imgs:
        for sub in (0, 1, 2):
            data = jpeg_bytes(im, quality=97, subsampling=sub)
            st, slot = entropy_decode(lib, data)
            assert st == 0
            assert np.array_equal(oj.decode_slot(slot), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def test_files_outside_the_fast_path_are_refused():
    lib = host_lib()
    rng = np.random.default_rng(1)
    im = synth_image(rng, 120, 160)
    assert entropy_decode(lib, jpeg_bytes(im, quality=80, progressive=True))[0] == 1                # progressive
    assert entropy_decode(lib, jpeg_bytes(im.convert("CMYK"), quality=80))[0] == 1                  # four components
    assert entropy_decode(lib, jpeg_bytes(synth_image(rng, 12, 12), quality=80))[0] == 1            # below 16 x 16
    buf = io.BytesIO()
    im.save(buf, "PNG")
    arr = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    slot = np.zeros(1 << 20, dtype=np.uint8)
    assert lib.hipts_jpeg_entropy_decode(arr.ctypes.data, len(arr), slot.ctypes.data, len(slot)) == 1    # not a JPEG
    data = jpeg_bytes(im, quality=80)
    assert entropy_decode(lib, data, slot_bytes=4096)[0] == 2                                        # slot too small
    for cut in (len(data) // 2, len(data) - 40, 300):
        assert entropy_decode(lib, data[:cut], slot_bytes=lib.hipts_jpeg_slot_bytes(160, 120))[0] == 3    # truncated: Pillow's business
    broken = bytearray(jpeg_bytes(im, quality=80, restart_marker_blocks=5))
    i = broken.index(b"\xff\xd3")
    broken[i + 1] = 0xD5                                                                            # restart markers out of sequence
    assert entropy_decode(lib, bytes(broken), slot_bytes=lib.hipts_jpeg_slot_bytes(160, 120))[0] == 3
