#!/usr/bin/env python3
"""Development aid: the attention kernel alone on the chip, ViT-B/16 @448 sub-batch shape (gpurun only).
usage: attn_time.py [tokens [batch]]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
from hiptagsearch import _lib
lib = _lib.load()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 784
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, TP, HD = 12, (T + 63) // 64 * 64, 64
rng = np.random.default_rng(0)
def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
q = np.zeros((B * H, TP, HD), np.float32); k = np.zeros_like(q); v = np.zeros((B * H, HD, TP), np.float32)
q[:, :T] = rng.standard_normal((B * H, T, HD)) * 0.18 * 1.4427      # head_dim^-0.5 * log2 e folded, |S| of a few units
k[:, :T] = rng.standard_normal((B * H, T, HD)); v[:, :, :T] = rng.standard_normal((B * H, HD, T))
qb, kb, vb = bf16(q), bf16(k), bf16(v)
fn = lib.hiptsdbg_attention_time
fn.restype = ctypes.c_int
flop = 4.0 * T * T * HD * B * H
for rep in range(3):
    us = ctypes.c_double()
    rc = fn(qb.ctypes.data_as(ctypes.c_void_p), kb.ctypes.data_as(ctypes.c_void_p), vb.ctypes.data_as(ctypes.c_void_p), B, H, T, TP, HD, 0, 30, ctypes.byref(us))
    assert rc == 0, rc
    print("attention %d x %d heads x %d tokens: %.1f us per launch  %.0f TFLOP/s" % (B, H, T, us.value, flop / us.value / 1e6), flush=True)
