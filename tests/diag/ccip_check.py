#!/usr/bin/env python3
"""Development aid: CCIP encoder vs oracle, bf16 and half operands (gpurun only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import synth
from hiptagsearch.cfeatures import CCIPEncoder
from oracle import ccip as oc
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = dict(synth.CCIP_TINY if which == "tiny" else synth.CCIP_B36_384)
n = 4 if which == "tiny" else 2
w = synth.ccip_weights(cfg, seed=3)
imgs = synth.images_u8(n, cfg["image_size"], seed=47)
x = oc.preprocess_u8_nhwc(imgs)
t = time.time()
want = oc.metaformer_forward(oc.to_torch(w), x, dims=cfg["dims"], depths=cfg["depths"]).numpy()
want8 = oc.metaformer_forward(oc.to_torch(w), x, dims=cfg["dims"], depths=cfg["depths"], e4m3=True).numpy()
print("oracle %.1f s" % (time.time() - t))
cos8 = (want8 * want).sum(1) / (np.linalg.norm(want8, axis=1) * np.linalg.norm(want, axis=1))
print("e4m3-emulating oracle vs float32 oracle: max|df| %.3e  rms %.3e  min cos %.6f" % (np.abs(want8 - want).max(), np.sqrt(((want8 - want) ** 2).mean()), cos8.min()))
for f16 in (0, 1, 2):
    c = dict(cfg); c["operand_f16"] = f16
    enc = CCIPEncoder(c, w, max_batch=n)
    got = enc.forward_u8(imgs)
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    print("operand_f16=%d  max|df| %.3e  rms %.3e  min cos %.6f" % (f16, np.abs(got - want).max(), np.sqrt(((got - want) ** 2).mean()), cos.min()))
    if f16 == 2:
        cos = (got * want8).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want8, axis=1))
        print("   vs the e4m3-emulating oracle: max|df| %.3e  rms %.3e  min cos %.6f" % (np.abs(got - want8).max(), np.sqrt(((got - want8) ** 2).mean()), cos.min()))
