#!/usr/bin/env python3
"""Development aid (gpurun only; csrc/mlp.hip built with -DHIPTS_MLP_STAMPS=<workgroup>): phases of one chunk of the fused MLP kernel, wave 0
of that workgroup, in microseconds."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_mlp_fused
f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
rng = np.random.default_rng(0)
for M, C in ((32 * 96 * 96, 128), (32 * 48 * 48, 256)):
    xn = rng.standard_normal((M, C)).astype(np.float16)
    w1 = (rng.standard_normal((4 * C, C)) / np.sqrt(C)).astype(np.float32)
    w2 = (rng.standard_normal((C, 4 * C)) / np.sqrt(4 * C)).astype(np.float32)
    x = rng.standard_normal((M, C)).astype(np.float32)
    rs = np.ones(C, dtype=np.float32); g = np.ones(C, dtype=np.float32)
    xo = np.zeros((M, C), dtype=np.float16)
    ms = ctypes.c_float(0)
    assert f(xn.ctypes.data, w1.ctypes.data, w2.ctypes.data, x.ctypes.data, rs.ctypes.data, g.ctypes.data, xo.ctypes.data, M, C, 0.8944, -0.4472, 1e-6, 4, ctypes.byref(ms), 8) == 0, _lib.last_error()
    st = (ctypes.c_ulonglong * 16)()
    assert lib.hiptsdbg_mlp_stamps(st, 16) == 0
    t = np.array(list(st), dtype=np.float64) / 100.0
    names = ["copy requested", "first product", "StarReLU", "second product", "own copies landed", "barrier"]
    print("M %d C %d: launch %.1f us; wave lifetime %.2f us = start to last chunk %.2f + epilogue %.2f; chunk 8: " % (M, C, ms.value * 1e3, t[10] - t[8], t[9] - t[8], t[10] - t[9])
          + ", ".join("%s %.2f" % (n, t[i + 1] - t[i]) for i, n in enumerate(names)))
