#!/usr/bin/env python3
"""Development aid: ViT-B/16 logit error vs the fp32 oracle for a few weight / image seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
torch.set_num_threads(16)
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger
from oracle import vit as ovit
cfg = dict(synth.VIT_B16_448)
for ws, iseed in [(1, 7), (2, 8), (3, 9)]:
    w = synth.vit_weights(cfg, seed=ws)
    imgs = synth.images_u8(4, 448, seed=iseed)
    imgs[1] = (imgs[1] // 64) * 64            # blocky image
    imgs[2, :, :, :] = imgs[2, :1, :1, :]     # constant-colour image
    want = ovit.vit_forward(ovit.to_torch(w), ovit.preprocess_u8_nhwc(imgs)).numpy()
    for f16 in (0, 1):
        c2 = dict(cfg); c2["operand_f16"] = f16
        model = ViTTagger(c2, w, max_batch=4)
        got, _ = model.forward_u8(imgs)
        print("%s weights seed %d images seed %d: max |dlogit| per image (random, blocky, flat, random) %s  (logit rms %.3f)" % (
            "f16 " if f16 else "bf16", ws, iseed, np.abs(got - want).max(axis=1), np.sqrt((want ** 2).mean())), flush=True)
        model.close()
