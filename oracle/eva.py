"""EVA02 tagger forward oracle (torch CPU, float32) -- test infrastructure, see oracle/__init__.py.

PARITY UNPINNED: this is the model the reference really loads (`SmilingWolf/wd-eva02-large-tagger-v3`,
tagging.py:45,146-148) through timm 1.0.9 `models/eva.py`; neither timm nor the weights are in this
container, so the graph is restated from the published definition of `eva02_large_patch14_448`
(SURVEY.md A2 / f1): patch 14 -> 32 x 32 patches + class token, absolute pos_embed, 24 blocks of
  x = x + proj(attn(rope(q), rope(k), v))          q/k/v separate Linear (q, v with bias, k without),
                                                    16 heads x 64, 2-D axial RoPE on the patch tokens only
  x = x + fc2(LN(silu(fc1_g(x')) * fc1_x(x')))      SwiGLU with an inner LayerNorm (scale_mlp), hidden 2730
with pre-LayerNorms (eps 1e-6), then mean over the PATCH tokens -> fc_norm (LayerNorm) -> head.
RoPE is timm's `RotaryEmbeddingCat(head_dim, in_pixels=False, feat_shape=grid, ref_feat_shape=(16, 16))`:
bands 1 / 10000^(i/16), i < 16, positions rescaled to the 16 x 16 reference grid, [y bands | x bands]
each repeated twice, applied as x * cos + rot(x) * sin with rot(x) = (-x_odd, x_even) interleaved.
The reference side anchor is the call contract (tagging.py:164,174,176).
"""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def rope_tables(grid: int, head_dim: int = 64, ref_grid: int = 16, temperature: float = 10000.0):
    """(sin, cos) float32 [grid*grid, head_dim]."""
    nb = head_dim // 4
    bands = 1.0 / (temperature ** (torch.arange(0, nb, dtype=torch.float32) / nb))
    t = torch.arange(grid, dtype=torch.float32) / grid * ref_grid
    gy, gx = torch.meshgrid(t, t, indexing="ij")
    pos = torch.stack([gy, gx], dim=-1).unsqueeze(-1) * bands                  # [H, W, 2, nb]
    sin = pos.sin().reshape(grid * grid, -1).repeat_interleave(2, dim=-1)
    cos = pos.cos().reshape(grid * grid, -1).repeat_interleave(2, dim=-1)
    return sin, cos


def _rot(x: torch.Tensor) -> torch.Tensor:
    return torch.stack([-x[..., 1::2], x[..., ::2]], dim=-1).reshape(x.shape)


@torch.no_grad()
def eva_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, *, patch: int = 14, heads: int = 16, eps: float = 1e-6,
                ref_grid: int = 16) -> torch.Tensor:
    """x: float32 [B,3,S,S] (normalised, BGR as tagging.py:243 hands it over).  Returns logits [B, num_classes]."""
    B = x.shape[0]
    t = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    grid = int(round(math.sqrt(t.shape[1])))
    t = torch.cat([w["cls_token"].expand(B, -1, -1), t], dim=1) + w["pos_embed"]
    N, D = t.shape[1], t.shape[2]
    hd = D // heads
    sin, cos = rope_tables(grid, hd, ref_grid)
    depth = 1 + max(int(k.split(".")[1]) for k in w if k.startswith("blocks."))
    for i in range(depth):
        p = "blocks.%d." % i
        h = F.layer_norm(t, (D,), w[p + "norm1.weight"], w[p + "norm1.bias"], eps)
        q = F.linear(h, w[p + "attn.q_proj.weight"], w[p + "attn.q_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
        k = F.linear(h, w[p + "attn.k_proj.weight"]).reshape(B, N, heads, hd).transpose(1, 2)
        v = F.linear(h, w[p + "attn.v_proj.weight"], w[p + "attn.v_proj.bias"]).reshape(B, N, heads, hd).transpose(1, 2)
        q = torch.cat([q[:, :, :1], q[:, :, 1:] * cos + _rot(q[:, :, 1:]) * sin], dim=2)
        k = torch.cat([k[:, :, :1], k[:, :, 1:] * cos + _rot(k[:, :, 1:]) * sin], dim=2)
        a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(o, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        h = F.layer_norm(t, (D,), w[p + "norm2.weight"], w[p + "norm2.bias"], eps)
        g = F.linear(h, w[p + "mlp.fc1_g.weight"], w[p + "mlp.fc1_g.bias"])
        u = F.linear(h, w[p + "mlp.fc1_x.weight"], w[p + "mlp.fc1_x.bias"])
        m = F.silu(g) * u
        m = F.layer_norm(m, (m.shape[-1],), w[p + "mlp.norm.weight"], w[p + "mlp.norm.bias"], eps)
        t = t + F.linear(m, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    f = t[:, 1:].mean(dim=1)                                                   # global_pool='avg' over the patch tokens
    f = F.layer_norm(f, (D,), w["fc_norm.weight"], w["fc_norm.bias"], eps)
    return F.linear(f, w["head.weight"], w["head.bias"])


def to_torch(weights: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
