#!/usr/bin/env python3
"""Can e4m3 operands carry the CCIP encoder?  CPU-only emulation (torch), development aid for the fp8 decision of round 4.

oracle/ccip.py's forward with the operands of pwconv2 / fc1 / fc2 rounded to OCP e4m3 in four ways, feature cosine against the
float32 forward on the B36 @384 model (seeded weights, two noise images + two structured ones):
  tensor      what operand_f16 = 2 does today: activations as they are (saturating), one power-of-two scale per weight matrix
  block       MX-style: one E8M0 (power-of-two) scale per 32 consecutive K elements, activations AND weights -- the scales the
              v_mfma_scale_f32_16x16x128_f8f6f4 instruction takes
  block23     the same in stages 2-3 only (the MFMA-bound stages), stages 0-1 in float32
  block_w16   block-scaled e4m3 activations against unrounded weights (isolates the activation side)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util

import numpy as np
import torch

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "anime-illust-image-searcher_amd", "hiptagsearch", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
from oracle import ccip as occip


def q8_block(a: torch.Tensor) -> torch.Tensor:
    """per 32 consecutive elements of the last axis: scale by the power of two that puts the block's max in (224, 448], round to e4m3, undo"""
    sh = a.shape
    K = sh[-1]
    assert K % 32 == 0
    b = a.reshape(-1, K // 32, 32)
    mx = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(448.0 / mx))
    e = torch.where(mx * torch.exp2(e) > 448.0, e - 1, e)
    s = torch.exp2(e)
    return ((b * s).to(torch.float8_e4m3fn).to(torch.float32) / s).reshape(sh)


def main():
    torch.set_num_threads(8)
    cfg = dict(synth.CCIP_B36_384)
    w = occip.to_torch(synth.ccip_weights(cfg, seed=46))
    imgs = np.concatenate([synth.images_u8(2, 384, seed=47), synth.structured_images_u8(384, seed=77, kinds=("lineart", "blocks"))])
    x = occip.preprocess_u8_nhwc(imgs)
    kw = dict(dims=cfg["dims"], depths=cfg["depths"], head_dim=cfg["head_dim"], eps=cfg["ln_eps"])
    t0 = time.time()
    ref = occip.metaformer_forward(w, x, **kw)
    print("float32 forward of %d images: %.1f s" % (len(imgs), time.time() - t0), flush=True)

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a, b, dim=1).numpy()

    ident = lambda t: t
    modes = {"tensor": (occip._q8, occip._q8w, None), "block": (q8_block, q8_block, None), "block23": (q8_block, q8_block, (2, 3)),
             "block_w16": (q8_block, ident, None)}
    orig = (occip._q8, occip._q8w)
    for name, (qa, qw, stages) in modes.items():
        occip._q8, occip._q8w = qa, qw
        if stages is None:
            got = occip.metaformer_forward(w, x, e4m3=True, **kw)
        else:       # only the named stages: run with a dims-dependent switch by zeroing the rounding elsewhere
            dims = cfg["dims"]
            qa0, qw0 = qa, qw
            keep = {dims[s] for s in stages}
            occip._q8 = lambda t, qa0=qa0, keep=keep: qa0(t) if (t.shape[-1] in keep or t.shape[-1] // 2 in keep or t.shape[-1] // 4 in keep) and t.shape[-1] >= min(keep) else t
            occip._q8w = lambda m, qw0=qw0, keep=keep: qw0(m) if min(m.shape) in keep else m
            got = occip.metaformer_forward(w, x, e4m3=True, **kw)
        c = cos(got, ref)
        print("%-10s cosine per image %s   min %.5f   max |df| %.3f" % (name, np.array2string(c, precision=5), c.min(), float((got - ref).abs().max())), flush=True)
    occip._q8, occip._q8w = orig


if __name__ == "__main__":
    main()
