#!/usr/bin/env python3
"""Development aid: EVA02 forward vs oracle, bf16 and half operands (gpurun only)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import synth
from hiptagsearch.tagger import EvaTagger
from oracle import eva as oe, vit as ovit
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = dict(synth.EVA02_TINY if which == "tiny" else synth.EVA02_L14_448)
n = 4 if which == "tiny" else 2
w = synth.eva_weights(cfg, seed=1)
imgs = synth.images_u8(n, cfg["image_size"], seed=2)
x = ovit.preprocess_u8_nhwc(imgs)
want = oe.eva_forward(oe.to_torch(w), x, patch=cfg["patch"], heads=cfg["heads"]).numpy()
for f16 in (0, 1):
    c = dict(cfg); c["operand_f16"] = f16
    m = EvaTagger(c, w, max_batch=n)
    got, _ = m.forward_u8(imgs)
    print("operand_f16=%d  max|dlogit| %.3e  rms %.3e (logit rms %.3f)" % (f16, np.abs(got - want).max(), np.sqrt(((got - want) ** 2).mean()), np.sqrt((want ** 2).mean())))
