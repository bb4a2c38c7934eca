#!/usr/bin/env python3
"""Development aid: the fused MLP kernel of the CCIP encoder's stages 0-1 alone (hiptsdbg_mlp_fused) at the shapes of a 32-image sub-batch
of CAFormer-B36 @384 (gpurun only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_mlp_fused
f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(0)
for M, C in ((32 * 96 * 96, 128), (32 * 48 * 48, 256), (10 * 96 * 96, 128), (10 * 48 * 48, 256), (20 * 48 * 48, 256)):
    xn = rng.standard_normal((M, C)).astype(np.float16)
    w1 = (rng.standard_normal((4 * C, C)) / np.sqrt(C)).astype(np.float32)
    w2 = (rng.standard_normal((C, 4 * C)) / np.sqrt(4 * C)).astype(np.float32)
    x = rng.standard_normal((M, C)).astype(np.float32)
    rs = np.ones(C, dtype=np.float32); g = np.ones(C, dtype=np.float32)
    xo = np.zeros((M, C), dtype=np.float16)
    fl = 2.0 * M * C * 4 * C * 2
    mb = M * C * (2 + 8 + 2) / 1e6
    for waves in (8, 4):
      ms = ctypes.c_float(0)
      assert f(xn.ctypes.data, w1.ctypes.data, w2.ctypes.data, x.ctypes.data, rs.ctypes.data, g.ctypes.data, xo.ctypes.data, M, C, 0.8944, -0.4472, 1e-6, 11, ctypes.byref(ms), waves) == 0, _lib.last_error()
      print("waves %d  M %d C %d: %.1f us  %.0f TFLOP/s  %.2f TB/s of the %.0f MB (xn in, x in + out, xn out)" % (waves, M, C, ms.value * 1e3, fl / ms.value / 1e9, mb / ms.value / 1e3, mb), flush=True)
