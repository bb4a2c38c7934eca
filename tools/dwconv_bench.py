#!/usr/bin/env python3
"""Development aid: the depthwise 7x7 of the CCIP encoder alone (hiptsdbg_dwconv7), float32-FMA kernel vs the matrix-core kernel,
at the shapes of a 32-image sub-batch of CAFormer-B36 @384 (gpurun only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_dwconv7
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
rng = np.random.default_rng(0)
for H, C in ((96, 256), (48, 512)):
    B = int(os.environ.get("B", "32"))
    x = rng.standard_normal((B, H, H, C)).astype(np.float16)
    w = (rng.standard_normal((C, 49)) * 0.2).astype(np.float32)
    out = np.empty_like(x)
    mb = x.nbytes * 2 / 1e6
    ref = None
    for mode in [int(a) for a in os.environ.get("MODES", "0,2,3").split(",")]:
        ms = ctypes.c_float(0)
        assert f(x.ctypes.data, w.ctypes.data, out.ctypes.data, B, H, C, mode, 20, ctypes.byref(ms)) == 0, _lib.last_error()
        if ref is None: ref = out.copy()
        d = np.abs(out.astype(np.float32) - ref.astype(np.float32)).max()
        print("H %d C %d mode %d: %.1f us  %.2f TB/s of the %.0f MB in + out   max |diff to mode 0| %.3g" % (H, C, mode, ms.value * 1e3, mb / ms.value / 1e3, mb, d), flush=True)
