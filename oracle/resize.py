"""PIL's 8-bit resample restated in numpy -- test infrastructure, see oracle/__init__.py.

Follows Pillow libImaging/Resample.c (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
Vertical_8bpc, ImagingResample): the algorithm behind `image.resize(size, BICUBIC)` of the timm eval transform
(tagging.py:241 on the padded square of tagging.py:100-120) and `image.resize((384, 384), BILINEAR)` (gen_cfeatures.py:101).
Pinned by tests/test_oracle_resize.py against Pillow itself (installed in the build image) and by the digests of
tests/golden/g10_resize.json.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = 2, 3


def _filter(x: float, kind: int) -> float:
    if x < 0.0:
        x = -x
    if kind == BICUBIC:
        a = -0.5
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size: int, out_size: int, kind: int):
    """(ksize, bounds int [out,2], kk int [out,ksize]) for the whole-image box."""
    filterscale = scale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = (2.0 if kind == BICUBIC else 1.0) * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_filter((x + xmin - center + 0.5) * ss, kind) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, xmax)
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
    return ksize, bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_u8(img: np.ndarray, out_h: int, out_w: int, kind: int) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [out_h,out_w,3], bit for bit Image.resize((out_w, out_h), kind)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    need_h, need_v = out_w != W, out_h != H
    cur = img
    y0 = 0
    if need_v:
        _, bv, kv = precompute_coeffs(H, out_h, kind)
        y0, y1 = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
    else:
        y1 = H
    if need_h:
        _, bh, kh = precompute_coeffs(W, out_w, kind)
        rows = cur[y0:y1].astype(np.int64)                           # the horizontal pass touches only the rows the vertical one reads
        tmp = np.empty((y1 - y0, out_w, 3), dtype=np.uint8)
        for xx in range(out_w):
            xmin, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(rows[:, xmin:xmin + n, :], kh[xx, :n], axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    else:
        cur = cur[y0:y1]
    if need_v:
        out = np.empty((out_h, cur.shape[1], 3), dtype=np.uint8)
        c64 = cur.astype(np.int64)
        for yy in range(out_h):
            ymin, n = int(bv[yy, 0]) - y0, int(bv[yy, 1])
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kv[yy, :n], c64[ymin:ymin + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
        cur = out
    return np.ascontiguousarray(cur)
