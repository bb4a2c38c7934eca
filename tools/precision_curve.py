#!/usr/bin/env python3
"""Logit error against the float32 oracle vs images/s for the operand modes of the ViT tagger (GPU box).

For each mode of hipts_vit_config_t.operand_f16 (0 bf16, 1 half, 17 half + attention output as a hi | lo pair, 16 bf16 + that
pair) the trained-like checkpoint tags the eight check images of bench.py (six structured kinds + two noise images) and the
table gives max |dlogit| per kind; then the same model is timed at batch 64 on device-resident input (median of `--rounds`
interleaved rounds of `--steps` forwards each, all modes in ONE process: cdna_hip_programming.md rule 24).
Writes one JSON object (stdout) -- the measured points of the cost curve in DESIGN.md section 2.

    python tools/precision_curve.py [--eva] > gpurun_out/precision_curve.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="1,17,0,16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--eva", action="store_true", help="EVA02-L/14 (batch 10) instead of ViT-B/16")
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    from hiptagsearch import synth
    from hiptagsearch.tagger import EvaTagger, ViTTagger
    torch.set_num_threads(a.threads)
    modes = [int(m) for m in a.modes.split(",")]
    from oracle import vit as ovit
    if a.eva:
        from oracle import eva as oeva
        cfg = dict(synth.EVA02_L14_448)
        w = synth.eva_weights(cfg, seed=0, trained_like=True)
        batch = min(a.batch, 10)
    else:
        cfg = dict(synth.VIT_B16_448)
        w = synth.vit_weights(cfg, seed=0, trained_like=True)
        batch = a.batch
    S = cfg["image_size"]
    chk = np.concatenate([synth.structured_images_u8(S, seed=77), synth.images_u8(2, S, seed=5)])
    kinds = list(synth.STRUCTURED_KINDS) + ["noise", "noise"]
    t0 = time.time()
    if a.eva:
        want = oeva.eva_forward(oeva.to_torch(w), ovit.preprocess_u8_nhwc(chk), patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"],
                                ref_grid=cfg["rope_ref_grid"]).numpy()
    else:
        want = ovit.vit_forward(ovit.to_torch(w), ovit.preprocess_u8_nhwc(chk), patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"],
                                gelu_kind="tanh" if cfg["gelu_tanh"] else "erf").numpy()
    print("oracle: %d images in %.1f s, logit rms %.3f" % (len(chk), time.time() - t0, np.sqrt((want ** 2).mean())), file=sys.stderr, flush=True)
    imgs = torch.from_numpy(synth.images_u8(batch, S, seed=99)).cuda()
    logits = torch.empty((batch, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    probs = torch.empty_like(logits)
    models, out = {}, {"model": "eva02_large" if a.eva else "vit_b16", "batch": batch, "logit_rms": float(np.sqrt((want ** 2).mean())),
                       "kinds": kinds, "modes": {}}
    for m in modes:
        model = (EvaTagger if a.eva else ViTTagger)(dict(cfg, operand_f16=m), w, max_batch=max(batch, len(chk)))
        got, _ = model.forward_u8(chk)
        d = got.astype(np.float64) - want.astype(np.float64)
        rec = {"max_abs": [float(v) for v in np.abs(d).max(axis=1)], "rms": [float(v) for v in np.sqrt((d ** 2).mean(axis=1))]}
        print("mode %2d  max |dlogit| " % m + " ".join("%s %.2e" % (k[:5], v) for k, v in zip(kinds, rec["max_abs"])), file=sys.stderr, flush=True)
        out["modes"][str(m)] = rec
        models[m] = model
        for _ in range(3):
            model.forward_u8(imgs, logits, probs)
        torch.cuda.synchronize()
    times = {m: [] for m in modes}
    for _ in range(a.rounds):
        for m in modes:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                models[m].forward_u8(imgs, logits, probs)
            torch.cuda.synchronize()
            times[m].append((time.perf_counter() - t0) / a.steps)
    for m in modes:
        med = float(np.median(times[m]))
        out["modes"][str(m)]["images_per_s_median"] = batch / med
        out["modes"][str(m)]["images_per_s_best"] = batch / min(times[m])
        print("mode %2d  %.0f images/s (median of %d rounds), best %.0f" % (m, batch / med, a.rounds, batch / min(times[m])), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
