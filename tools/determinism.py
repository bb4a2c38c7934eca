#!/usr/bin/env python3
"""Development aid: run the same forward several times and report which workspace buffer first differs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import synth, _lib
from hiptagsearch.tagger import ViTTagger
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dict(synth.VIT_B16_448); cfg["depth"] = depth
w = synth.vit_weights(cfg, seed=5)
imgs = synth.images_u8(B, 448, seed=6)
model = ViTTagger(cfg, w, max_batch=B)
lib = _lib.load()
f = lib.hiptsdbg_vit_dump
f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
names = ["a0", "q", "k", "v", "att", "xn", "hmid", "x", "pool_part", "pooled2"]
def snap():
    logits, _ = model.forward_u8(imgs)
    out = {"logits": logits.copy()}
    for n in names:
        buf = np.empty(64 << 20, dtype=np.uint8); nb = ctypes.c_size_t()
        assert f(model._h, n.encode(), buf.ctypes.data, buf.nbytes, ctypes.byref(nb)) == 0, _lib.last_error()
        out[n] = buf[:nb.value].copy()
    return out
ref = snap()
for it in range(3):
    cur = snap()
    print("run", it, {n: int((cur[n] != ref[n]).sum()) for n in names + ["logits"]}, flush=True)
a = ref["att"].view(np.uint16).reshape(-1, 768)[: B * 784]
for it in range(2):
    c = snap()["att"].view(np.uint16).reshape(-1, 768)[: B * 784]
    rows, cols = np.nonzero(a != c)
    toks = rows % 784
    print("att diffs: n=%d  tokens min/max %d %d  unique q-tiles(32) %s  heads %s  d%%64 range %d-%d" % (
        len(rows), toks.min() if len(rows) else -1, toks.max() if len(rows) else -1, sorted(set((toks // 32).tolist()))[:30],
        sorted(set((cols // 64).tolist())), (cols % 64).min() if len(rows) else -1, (cols % 64).max() if len(rows) else -1), flush=True)
    af = (a.astype(np.uint32) << 16).view(np.float32); cf = (c.astype(np.uint32) << 16).view(np.float32)
    print("   max abs diff %.3e, typical |att| %.3e" % (np.abs(af - cf).max(), np.abs(af).mean()))
