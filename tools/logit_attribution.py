#!/usr/bin/env python3
"""Which 16-bit operand carries the ViT logit error?  CPU-only attribution (torch fp32), development aid.

Restates the ViT forward with the HIP path's operand roundings as switches -- every tensor that the product
stores as a 16-bit MFMA operand (csrc/vit.hip forward: xn1 = 16bit(gamma1 * x) into q|k|v, q / k / v themselves,
P = 2^(S - m) into P V, the attention output into proj, xn2 = 16bit(gamma2 * x) into fc1, the GELU hidden tensor into
fc2) -- and reports the max |dlogit| against the unrounded forward per image kind, with ALL roundings on, with each
one alone, and with each one left out.  `--split NAME[,NAME]` models the hi|lo split of an operand (two 16-bit halves:
22 significant bits), `--blocks a-b` restricts the roundings to a range of blocks, `--bf16` uses bf16 roundings,
`--dither` models a per-element hash dither ahead of the rounding (decorrelates the error between tokens).

    python tools/logit_attribution.py                   # trained-like checkpoint, 2 noise + 6 structured images
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import importlib.util

import numpy as np
import torch
import torch.nn.functional as F

# hiptagsearch/__init__ loads the HIP library; the synthetic generators are plain numpy -- import the module file itself
_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "anime-illust-image-searcher_amd", "hiptagsearch", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)
from oracle import vit as ovit

POINTS = ("xn1", "qkv", "P", "att", "xn2", "hmid")


class Rounder:
    def __init__(self, on, split, bf16, blocks, dither):
        self.on, self.split, self.bf16, self.blocks, self.dither = set(on), set(split), bf16, blocks, dither
        self.gen = torch.Generator().manual_seed(1)

    def r16(self, t):
        if self.bf16:
            return t.to(torch.bfloat16).to(torch.float32)
        return t.to(torch.float16).to(torch.float32)

    def __call__(self, name, t, block):
        if name not in self.on or not (self.blocks[0] <= block <= self.blocks[1]):
            return t
        if name in self.split:                       # hi + lo halves, each 16 bits
            hi = self.r16(t)
            return hi + self.r16(t - hi)
        if self.dither:                              # uniform dither of one ulp ahead of round-to-nearest = stochastic rounding
            ulp = torch.exp2(torch.floor(torch.log2(t.abs().clamp_min(1e-30))) - (7 if self.bf16 else 10))
            t = t + (torch.rand(t.shape, generator=self.gen) - 0.5) * ulp
        return self.r16(t)


@torch.no_grad()
def forward(w, x, rd, heads=12, eps=1e-6, folded=True):
    t = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2) + w["pos_embed"]
    B, N, D = t.shape
    hd = D // heads
    depth = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("blocks."))

    def ln(t, g, b, name, i):
        mu = t.mean(-1, keepdim=True)
        rstd = torch.rsqrt(t.var(-1, unbiased=False, keepdim=True) + eps)
        if folded:        # the product's form: 16bit(gamma * x), mean / rstd / beta applied after the product
            return (rd(name, t * g, i) - mu * g) * rstd + b
        return rd(name, (t - mu) * rstd * g + b, i)

    for i in range(depth):
        p = "blocks.%d." % i
        h = ln(t, w[p + "norm1.weight"], w[p + "norm1.bias"], "xn1", i)
        qkv = F.linear(h, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q = rd("qkv", qkv[0] * (hd ** -0.5 * 1.4426950408889634), i)
        k = rd("qkv", qkv[1], i)
        v = rd("qkv", qkv[2], i)
        s = q @ k.transpose(-2, -1)
        pw = torch.exp2(s - s.max(-1, keepdim=True).values)
        l = pw.sum(-1, keepdim=True)
        o = ((rd("P", pw, i) @ v) / l).transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(rd("att", o, i), w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        h = ln(t, w[p + "norm2.weight"], w[p + "norm2.bias"], "xn2", i)
        h = ovit.gelu(F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]), "tanh")
        t = t + F.linear(rd("hmid", h, i), w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    f = F.layer_norm(t, (D,), w["norm.weight"], w["norm.bias"], eps).mean(dim=1)
    return F.linear(f, w["head.weight"], w["head.bias"])


@torch.no_grad()
def forward_eva(w, x, rd, heads=16, eps=1e-6, ref_grid=16, patch=14):
    """oracle/eva.py::eva_forward with the HIP path's operand roundings (csrc/eva.hip): xn1 = 16bit(gamma1 * x) into q | k | v; q, k (after
    the rotary embedding, q pre-scaled) and v; P; the attention output into proj; xn2 = 16bit(gamma2 * x) into the SwiGLU GEMM; hmid =
    16bit(gamma_mlp * silu(g) * u) into fc2 (the inner LayerNorm folded like the outer ones)."""
    from oracle import eva as oeva
    B = x.shape[0]
    t = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    grid = int(round(t.shape[1] ** 0.5))
    t = torch.cat([w["cls_token"].expand(B, -1, -1), t], dim=1) + w["pos_embed"]
    N, D = t.shape[1], t.shape[2]
    hd = D // heads
    sin, cos = oeva.rope_tables(grid, hd, ref_grid)
    depth = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("blocks."))

    def ln(t, g, b, name, i):
        mu = t.mean(-1, keepdim=True)
        rstd = torch.rsqrt(t.var(-1, unbiased=False, keepdim=True) + eps)
        return (rd(name, t * g, i) - mu * g) * rstd + b

    for i in range(depth):
        p = "blocks.%d." % i
        h = ln(t, w[p + "norm1.weight"], w[p + "norm1.bias"], "xn1", i)
        q = F.linear(h, w[p + "attn.q_proj.weight"], w[p + "attn.q_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
        k = F.linear(h, w[p + "attn.k_proj.weight"]).reshape(B, N, heads, hd).transpose(1, 2)
        v = F.linear(h, w[p + "attn.v_proj.weight"], w[p + "attn.v_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
        q = torch.cat([q[:, :, :1], q[:, :, 1:] * cos + oeva._rot(q[:, :, 1:]) * sin], dim=2)
        k = torch.cat([k[:, :, :1], k[:, :, 1:] * cos + oeva._rot(k[:, :, 1:]) * sin], dim=2)
        q = rd("qkv", q * (hd ** -0.5 * 1.4426950408889634), i)
        k = rd("qkv", k, i)
        v = rd("qkv", v, i)
        s = q @ k.transpose(-2, -1)
        pw = torch.exp2(s - s.max(-1, keepdim=True).values)
        l = pw.sum(-1, keepdim=True)
        o = ((rd("P", pw, i) @ v) / l).transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(rd("att", o, i), w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        h = ln(t, w[p + "norm2.weight"], w[p + "norm2.bias"], "xn2", i)
        g = F.linear(h, w[p + "mlp.fc1_g.weight"], w[p + "mlp.fc1_g.bias"])
        u = F.linear(h, w[p + "mlp.fc1_x.weight"], w[p + "mlp.fc1_x.bias"])
        m = ln(F.silu(g) * u, w[p + "mlp.norm.weight"], w[p + "mlp.norm.bias"], "hmid", i)
        t = t + F.linear(m, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    f = t[:, 1:].mean(dim=1)
    f = F.layer_norm(f, (D,), w["fc_norm.weight"], w["fc_norm.bias"], eps)
    return F.linear(f, w["head.weight"], w["head.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["vit", "eva"], default="vit", help="eva: EVA02-L/14, the model the reference loads (tagging.py:45); 3 images")
    ap.add_argument("--random-init", action="store_true", help="the random-init checkpoint instead of the trained-like one")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--dither", action="store_true")
    ap.add_argument("--split", default="", help="operands modelled as hi|lo pairs (comma list of %s)" % (POINTS,))
    ap.add_argument("--blocks", default="0-99")
    ap.add_argument("--unfolded", action="store_true", help="round LayerNorm's output instead of gamma * x")
    ap.add_argument("--quick", action="store_true", help="all-on and each-alone only")
    ap.add_argument("--only-all", action="store_true", help="the all-roundings row only")
    ap.add_argument("--seed", type=int, default=77, help="seed of the structured images (bench.py uses 77 too)")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    if a.model == "eva":        # the images of tests/test_gpu_eva.py::test_eva02_large_trained_like_checkpoint
        cfg = dict(synth.EVA02_L14_448)
        w = ovit.to_torch(synth.eva_weights(cfg, seed=0, trained_like=not a.random_init))
        imgs = np.concatenate([synth.images_u8(1, 448, seed=5), synth.structured_images_u8(448, seed=a.seed, kinds=("flat", "blocks"))])
        kinds = ["noise", "flat", "blocks"]
    else:
        cfg = dict(synth.VIT_B16_448)
        w = ovit.to_torch(synth.vit_weights(cfg, seed=0, trained_like=not a.random_init))
        imgs = np.concatenate([synth.images_u8(2, 448, seed=5), synth.structured_images_u8(448, seed=a.seed)])
        kinds = ["noise", "noise"] + list(synth.STRUCTURED_KINDS)
    x = ovit.preprocess_u8_nhwc(imgs)
    blocks = tuple(int(v) for v in a.blocks.split("-"))
    split = [s for s in a.split.split(",") if s]

    def run(on):
        rd = Rounder(on, split, a.bf16, blocks, a.dither)
        if a.model == "eva":
            return forward_eva(w, x, rd).numpy().astype(np.float64)
        return forward(w, x, rd, folded=not a.unfolded).numpy().astype(np.float64)

    t0 = time.time()
    base = run(())
    print("logit rms %.3f; one forward of %d images %.1f s" % (np.sqrt((base ** 2).mean()), len(imgs), time.time() - t0), flush=True)
    print("%-22s" % "roundings" + "".join("%11s" % k for k in kinds))

    def report(label, on):
        d = np.abs(run(on) - base).max(axis=1)
        print("%-22s" % label + "".join("%11.2e" % v for v in d), flush=True)

    report("all", POINTS)
    if a.only_all:
        return
    for pnt in POINTS:
        report("only " + pnt, (pnt,))
    if not a.quick:
        for pnt in POINTS:
            report("all but " + pnt, tuple(q for q in POINTS if q != pnt))


if __name__ == "__main__":
    main()
