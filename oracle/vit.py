"""ViT tagger forward oracle (torch CPU, float32) -- test infrastructure, see oracle/__init__.py.

PARITY UNPINNED: the graph is timm 1.0.9 `VisionTransformer` (absent from /root/reference
and not installed); restated from its published definition for the wd-vit-tagger-v3
configuration (vit_base_patch16, img 448, class_token=False, global_pool='avg',
fc_norm=False, LayerNorm eps 1e-6, qkv_bias=True) and anchored on the reference call
sites: tagging.py:241-243 (transform + BGR flip), :164 (stack), :174 (forward),
:176 (sigmoid).
The transformer blocks are additionally checked against HuggingFace transformers' ViTLayer
(an independent implementation, present in the image) by tests/test_oracle_vit_vs_transformers.py.

Input convention matches the product boundary: uint8 NHWC RGB images that are already
448x448 (resize/center-crop are the identity for them), so the timm eval transform is
ToTensor (/255) then Normalize(mean=.5, std=.5), followed by the channel flip.
"""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def preprocess_u8_nhwc(images_u8: np.ndarray) -> torch.Tensor:
    """[B,H,W,3] uint8 RGB -> float32 [B,3,H,W] BGR in [-1,1]  (tagging.py:241-243)."""
    x = torch.from_numpy(np.ascontiguousarray(images_u8)).permute(0, 3, 1, 2).to(torch.float32)
    x = x / 255.0                       # ToTensor
    x = (x - 0.5) / 0.5                 # Normalize(mean=.5, std=.5)
    return x[:, [2, 1, 0]]              # RGB -> BGR


def gelu(x: torch.Tensor, kind: str) -> torch.Tensor:
    return F.gelu(x, approximate="tanh") if kind == "tanh" else F.gelu(x)


@torch.no_grad()
def vit_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, *, patch: int = 16, heads: int = 12,
                eps: float = 1e-6, gelu_kind: str = "tanh", pool_then_norm: bool = False,
                return_tokens: bool = False) -> torch.Tensor:
    """x: float32 [B,3,H,W] (already normalised, BGR).  Returns logits [B,num_classes]."""
    B = x.shape[0]
    t = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=patch)   # PatchEmbed
    t = t.flatten(2).transpose(1, 2)                                                          # [B,N,D]
    t = t + w["pos_embed"]
    N, D = t.shape[1], t.shape[2]
    hd = D // heads
    depth = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("blocks."))
    for i in range(depth):
        p = "blocks.%d." % i
        h = F.layer_norm(t, (D,), w[p + "norm1.weight"], w[p + "norm1.bias"], eps)
        qkv = F.linear(h, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = (q * (hd ** -0.5)) @ k.transpose(-2, -1)
        a = a.softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(o, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        h = F.layer_norm(t, (D,), w[p + "norm2.weight"], w[p + "norm2.bias"], eps)
        h = gelu(F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]), gelu_kind)
        t = t + F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    if return_tokens:
        return t
    if pool_then_norm:      # timm fc_norm=True variant
        f = F.layer_norm(t.mean(dim=1), (D,), w["norm.weight"], w["norm.bias"], eps)
    else:                   # final norm on tokens, then average pool (fc_norm=False)
        f = F.layer_norm(t, (D,), w["norm.weight"], w["norm.bias"], eps).mean(dim=1)
    return F.linear(f, w["head.weight"], w["head.bias"])


def to_torch(weights: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
