#!/usr/bin/env python3
"""Development aid (gpurun only): every GEMM launch of the ViT-B/16 forward through the 8-wave loop (csrc/gemm.hip) and the 4-wave loop
(csrc/gemm4.hip) on the same operands -- the outputs must agree byte for byte -- with the average launch time each way."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
from hiptagsearch import _lib
lib = _lib.load()
f = lib.hiptsdbg_gemm_q4_compare
f.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_float)]
EPI = {"qk": 1, "gelu": 4, "xg": 13}
shapes = [("gelu", 784 * 16, 1024, 128), ("xg", 784 * 16, 768, 256), ("qk", 784 * 16, 768, 128), ("gelu", 784 * 32, 3072, 128),
           ("gelu", 25088, 3072, 768), ("qk", 25088, 2304, 768), ("xg", 25088, 768, 768), ("xg", 25088, 768, 3072),
           ("gelu", 50176, 3072, 768), ("xg", 50176, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [(a.split(",")[0], int(a.split(",")[1]), int(a.split(",")[2]), int(a.split(",")[3])) for a in sys.argv[1:]]
iters = int(os.environ.get("ITERS", "10"))
bad_total = 0
for f16 in (1, 0):
    for name, M, N, K in shapes:
        bad = ctypes.c_longlong()
        ms = (ctypes.c_float * 2)()
        st = f(M, N, K, EPI[name], f16, iters, ctypes.byref(bad), ms)
        if st:
            print(name, M, N, K, "error", _lib.last_error(), flush=True)
            bad_total += 1
            continue
        fl = 2.0 * M * N * K
        print("%-5s %s M=%6d N=%5d K=%5d  mismatching bytes %d   8-wave %8.1f us %7.1f TF   4-wave %8.1f us %7.1f TF   (x%.3f)" % (
            name, "f16 " if f16 else "bf16", M, N, K, bad.value, ms[0] * 1e3, fl / ms[0] / 1e9, ms[1] * 1e3, fl / ms[1] / 1e9, ms[0] / ms[1]), flush=True)
        bad_total += bad.value != 0
sys.exit(1 if bad_total else 0)
