#!/usr/bin/env python3
"""Development aid (gpurun only; csrc/ccip.hip built with -DHIPTS_DW_STAMPS=<workgroup>): phases of one workgroup of the matrix-core
depthwise 7x7 (dwconv7_mfma_kernel), in microseconds."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_dwconv7
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
rng = np.random.default_rng(0)
for H, C, mode in ((96, 256, 3), (48, 512, 3), (96, 256, 2)):
    B = 32
    x = rng.standard_normal((B, H, H, C)).astype(np.float16)
    w = (rng.standard_normal((C, 49)) * 0.2).astype(np.float32)
    out = np.empty_like(x)
    ms = ctypes.c_float(0)
    assert f(x.ctypes.data, w.ctypes.data, out.ctypes.data, B, H, C, mode, 3, ctypes.byref(ms)) == 0, _lib.last_error()
    st = (ctypes.c_ulonglong * 32)()
    assert lib.hiptsdbg_dwconv7_stamps(st, 32) == 0
    t = np.array(list(st), dtype=np.float64) / 100.0
    names = ["requested", "planes written", "barrier", "products", "barrier", "results in LDS", "stores issued"]
    print("H %d C %d mode %d: launch %.1f us; operands built %.2f us; workgroup lifetime %.2f us" % (H, C, mode, ms.value * 1e3, t[1] - t[0], t[31] - t[0]))
    for k in range(3):
        b = 2 + 8 * k
        if t[b] == 0: break
        print("   tile %d: " % k + ", ".join("%s %.2f" % (n, t[b + 1 + i] - t[b + i]) for i, n in enumerate(names)) + ", to next tile %.2f" % ((t[b + 8] if k < 2 and t[b + 8] else t[31]) - t[b + 7]))
