#!/usr/bin/env python3
"""Development aid: run one forward in half-operand mode and report NaN/inf per workspace buffer."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import synth, _lib
from hiptagsearch.tagger import ViTTagger
cfg = dict(synth.VIT_B16_448); cfg["depth"] = 1; cfg["operand_f16"] = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = synth.vit_weights(cfg, seed=5)
imgs = synth.images_u8(2, 448, seed=6)
model = ViTTagger(cfg, w, max_batch=2)
logits, _ = model.forward_u8(imgs)
lib = _lib.load(); f = lib.hiptsdbg_vit_dump
f.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
for n, dt in [("a0", "h"), ("q", "h"), ("k", "h"), ("v", "h"), ("att", "h"), ("xn", "h"), ("hmid", "h"), ("x", "f"), ("pool_part", "f"), ("pooled2", "h")]:
    buf = np.empty(64 << 20, dtype=np.uint8); nb = ctypes.c_size_t()
    assert f(model._h, n.encode(), buf.ctypes.data, buf.nbytes, ctypes.byref(nb)) == 0
    raw = buf[:nb.value]
    if dt == "f": v = raw.view(np.float32)
    elif cfg["operand_f16"]: v = raw.view(np.float16).astype(np.float32)
    else: v = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    print("%-10s n=%9d nan=%8d inf=%8d  absmax(finite)=%.4g" % (n, v.size, np.isnan(v).sum(), np.isinf(v).sum(), np.abs(v[np.isfinite(v)]).max() if np.isfinite(v).any() else -1), flush=True)
print("logits nan", np.isnan(logits).sum())
