"""Host-side weight layouts of the two round-4 CCIP kernels, without a GPU: the fused MLP's chunk images (csrc/mlp.hip mlp_weight_image)
and the lane images of the matrix-core depthwise 7x7's Toeplitz operands (csrc/ccip.hip dw_toeplitz_lanes), each rebuilt here in numpy
from the formulas in include/hip_tagsearch_debug.h.  The kernels that read them are checked against float64 in tests/test_gpu_ccip.py."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))


def _lib():
    from hiptagsearch import _lib
    return _lib.load(), _lib


@pytest.mark.parametrize("C", [128, 256])
def test_mlp_weight_image_layout(C):
    lib, L = _lib()
    rng = np.random.default_rng(C)
    hid = 4 * C
    w1 = rng.standard_normal((hid, C)).astype(np.float32)
    w2 = rng.standard_normal((C, hid)).astype(np.float32)
    p1, p2 = C + 16, 40                                   # row pitches in halves
    per = 32 * p1 + C * p2
    out = np.zeros((hid // 32) * per, dtype=np.uint16)
    f = lib.hiptsdbg_mlp_weight_image
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
    assert f(w1.ctypes.data, w2.ctypes.data, C, out.ctypes.data, out.size) == 0, L.last_error()
    want = np.zeros_like(out)
    h1, h2 = w1.astype(np.float16).view(np.uint16), w2.astype(np.float16).view(np.uint16)
    e = np.arange(8)
    for j in range(hid // 32):
        blk = want[j * per:(j + 1) * per]
        a = blk[:32 * p1].reshape(32, p1)
        a[:, :C] = h1[j * 32:(j + 1) * 32]
        b = blk[32 * p1:].reshape(C, p2)
        for q in range(4):
            b[:, 8 * q:8 * q + 8] = h2[:, j * 32 + 16 * (e >> 2) + 4 * q + (e & 3)]
    np.testing.assert_array_equal(out, want)
    assert f(w1.ctypes.data, w2.ctypes.data, C, out.ctypes.data, out.size - 1) != 0      # a wrong buffer size is refused
    assert f(w1.ctypes.data, w2.ctypes.data, 192, out.ctypes.data, out.size) != 0         # so is a width the kernel is not built for


def test_dw_toeplitz_lane_images():
    lib, L = _lib()
    rng = np.random.default_rng(7)
    ch = 48
    w = rng.standard_normal((ch, 49)).astype(np.float32)
    out = np.zeros((ch, 7, 64), dtype=np.uint32)
    f = lib.hiptsdbg_dw_toeplitz
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert f(w.ctypes.data, ch, out.ctypes.data) == 0, L.last_error()
    wh = w.astype(np.float16).view(np.uint16).reshape(ch, 7, 7)
    z = np.zeros((ch, 7, 50), dtype=np.uint32)
    z[:, :, 16:23] = wh
    want = np.zeros_like(out)
    l = np.arange(24)
    want[:, :, :24] = z[:, :, 2 * l] | (z[:, :, 2 * l + 1] << 16)
    want[:, :, 32:56] = z[:, :, 2 * l + 1] | (z[:, :, 2 * l + 2] << 16)
    np.testing.assert_array_equal(out, want)
    # what the kernel gathers from it: lane (x, q) of the second MFMA operand reads halves Z[s .. s + 7], s = 8 q - x + 15, i.e.
    # T[k][x] = w[ky][k - x - 1] for k = 8 q + i
    for x in (0, 5, 15):
        for q in range(4):
            s = 8 * q - x + 15
            tap = 8 * q + np.arange(8) - x - 1
            ok = (tap >= 0) & (tap < 7)
            exp = np.where(ok, wh[3, 2, np.clip(tap, 0, 6)], 0)
            np.testing.assert_array_equal(z[3, 2, s:s + 8].astype(np.uint16), exp)
