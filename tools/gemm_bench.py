#!/usr/bin/env python3
"""Development aid: time GEMM shapes through the internal hiptsdbg_gemm_time entry (gpurun only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_gemm_time
f.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
EPI = {"patch": 0, "qk": 1, "vt": 2, "resid": 3, "gelu": 4, "star": 6, "xg": 13}
shapes = [("qk", 50176, 1536, 768), ("vt", 50176, 768, 768), ("resid", 50176, 768, 768), ("gelu", 50176, 3072, 768),
          ("resid", 50176, 768, 3072), ("gelu", 4096, 4096, 4096), ("gelu", 8192, 8192, 8192), ("gelu", 50176, 3072, 3072)]
if len(sys.argv) > 1:
    shapes = [(a.split(",")[0], int(a.split(",")[1]), int(a.split(",")[2]), int(a.split(",")[3])) for a in sys.argv[1:]]
for name, M, N, K in shapes:
    ms = ctypes.c_float()
    st = f(M, N, K, EPI[name], 10, ctypes.byref(ms))
    if st:
        print(name, M, N, K, "error", _lib.last_error()); continue
    print("%-6s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, ms.value * 1e3, 2.0 * M * N * K / ms.value / 1e9), flush=True)
