#!/usr/bin/env python3
"""Development aid (gpurun only): does the ViT forward's rate depend on what the images show?  The attention kernel takes its fast path
(softmax against a fixed reference exponent) per workgroup and repeats a query block with the classic per-tile maximum when a row sum
leaves the window -- so the answer can be yes.  Trained-like checkpoint, batch 64, per kind of image content."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
import torch
from PIL import Image
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger

EVA = os.environ.get("MODEL") == "eva"          # MODEL=eva: EVA02-L/14 at the reference's batch of 10 x 2
N = 20 if EVA else 64
tl = os.environ.get("TRAINED_LIKE", "1") == "1"
if EVA:
    from hiptagsearch.tagger import EvaTagger
    cfg = dict(synth.EVA02_L14_448)
    model = EvaTagger(cfg, synth.eva_weights(cfg, 0, trained_like=tl), max_batch=N)
else:
    cfg = dict(synth.VIT_B16_448)
    model = ViTTagger(cfg, synth.vit_weights(cfg, seed=0, trained_like=tl), max_batch=N)
rng = np.random.default_rng(3)
sets = {"uniform noise (bench.py)": synth.images_u8(N, 448, seed=5)}
st = synth.structured_images_u8(448, seed=77)
for i, kind in enumerate(synth.STRUCTURED_KINDS):
    sets[kind + " x64"] = np.repeat(st[i:i + 1], N, axis=0)
sets["the six structured kinds mixed"] = np.concatenate([st] * 11)[:N]
sets = {k: v[:N] for k, v in sets.items()}
small = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).resize((1024, 768), Image.BICUBIC)
a = np.asarray(small, dtype=np.int16) + rng.integers(-6, 7, (768, 1024, 3), dtype=np.int16)
photo = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
sq = Image.new("RGB", (1024, 1024), (255, 255, 255)); sq.paste(photo, (0, 128))
sets["pipeline_e2e.py's picture (smooth + noise, padded) x64"] = np.repeat(np.asarray(sq.resize((448, 448), Image.BICUBIC))[None], N, axis=0)
def padded(w, h, noise=6, bg=None):
    small = Image.fromarray(rng.integers(0, 256, (max(2, h // 32), max(2, w // 32), 3), dtype=np.uint8)).resize((w, h), Image.BICUBIC)
    a = np.asarray(small, dtype=np.int16) + rng.integers(-noise, noise + 1, (h, w, 3), dtype=np.int16)
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    m = max(w, h)
    sq = Image.new("RGB", (m, m), (255, 255, 255))
    sq.paste(im, ((m - w) // 2, (m - h) // 2))
    return np.asarray(sq.resize((448, 448), Image.BICUBIC))


sets["portrait picture 600 x 1000, padded x64"] = np.repeat(padded(600, 1000)[None], N, axis=0)
sets["wide picture 1600 x 500, padded x64"] = np.repeat(padded(1600, 500)[None], N, axis=0)
obj = np.full((448, 448, 3), 245, np.uint8)
obj[150:300, 180:330] = padded(150, 150)[:150, :150]
sets["small object on a flat background x64"] = np.repeat(obj[None], N, axis=0)
sets["64 different padded pictures"] = np.stack([padded(int(rng.integers(400, 1600)), int(rng.integers(400, 1600))) for _ in range(N)])
probs = torch.empty((N, cfg["num_classes"]), dtype=torch.float32, device="cuda")
print("trained-like checkpoint" if tl else "random-init checkpoint")
for name, imgs in sets.items():
    d = torch.from_numpy(np.ascontiguousarray(imgs)).cuda()
    for _ in range(3):
        model.forward_u8(d, probs=probs, want="probs")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    REP = 6 if EVA else 20
    for _ in range(REP):
        model.forward_u8(d, probs=probs, want="probs")
    torch.cuda.synchronize()
    print("%-58s %6.0f images/s" % (name, N * REP / (time.perf_counter() - t0)), flush=True)
