"""CCIP feature-encoder oracle (torch CPU, float32) -- test infrastructure, see oracle/__init__.py.

PARITY UNPINNED: the reference runs an opaque ONNX graph (`deepghs/ccip_onnx/
ccip-caformer-24-randaug-pruned/model_feat.onnx`, gen_cfeatures.py:112-118,133-159) through
onnxruntime; neither the file, onnxruntime nor timm is in this container.  What the graph is known to be
(SURVEY.md A6): a CAFormer (timm 1.0.9 `models/metaformer.py`, `MetaFormer` with SepConv token mixers in
stages 1-2, self-attention with head_dim 32 in stages 3-4, StarReLU MLPs, bias-free LayerNorms,
res_scale in stages 3-4) at 384x384 whose pooled, normalised feature (768-d for the B36 widths) is the
output -- tensor names `input` -> `output` (gen_cfeatures.py:158).  This file restates that published
definition; the anchor on the reference side is the call contract: float32 [B,3,384,384] RGB normalised
with the CLIP mean / std (gen_cfeatures.py:100-110) in, float32 [B,768] out.
"""
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # gen_cfeatures.py:103-104
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_u8_nhwc(images_u8: np.ndarray) -> torch.Tensor:
    """[B,S,S,3] uint8 RGB (already S x S) -> float32 [B,3,S,S]: x/255, (x - mean) / std in float64, cast
    to float32 (gen_cfeatures.py:100-110: `_normalize` runs on float64 data, the cast is at :156)."""
    x = images_u8.astype(np.float32).transpose(0, 3, 1, 2) / 255.0                    # :107 float32 / 255
    mean = np.asarray(CLIP_MEAN, dtype=np.float64).reshape(1, 3, 1, 1)
    std = np.asarray(CLIP_STD, dtype=np.float64).reshape(1, 3, 1, 1)
    return torch.from_numpy(((x - mean) / std).astype(np.float32))


def star_relu(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    return scale.reshape(()) * torch.relu(x) ** 2 + bias.reshape(())


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _q8(a: torch.Tensor) -> torch.Tensor:
    """Activation as an e4m3 operand: saturate at +-448, round to nearest even (OCP e4m3fn)."""
    return a.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def _q8w(m: torch.Tensor) -> torch.Tensor:
    """Weight matrix as stored by the e4m3 mode: e4m3(W 2^e) 2^-e with the per-tensor power of two e that puts
    max |W| in (224, 448] (csrc/vit_internal.h upload_matrix8)."""
    mx = float(m.abs().max())
    if mx == 0.0:
        return m
    e = int(np.floor(np.log2(448.0 / mx)))
    if mx * 2.0 ** e > 448.0:
        e -= 1
    return (m * 2.0 ** e).to(torch.float8_e4m3fn).to(torch.float32) * 2.0 ** -e


@torch.no_grad()
def metaformer_forward(w: Dict[str, torch.Tensor], x: torch.Tensor, *, dims: Sequence[int], depths: Sequence[int],
                       head_dim: int = 32, eps: float = 1e-6, attn_from_stage: int = 2, e4m3: bool = False) -> torch.Tensor:
    """x: float32 [B,3,S,S] (normalised RGB).  Returns the pooled, normalised feature [B, dims[-1]].
    e4m3=True emulates the library's fp8 operand mode (hipts_ccip_config_t.operand_f16 = 2): in stages whose width is a
    multiple of 128 the operands of pwconv2, fc1 and fc2 -- activations and weights -- are rounded to e4m3 first;
    everything else stays float32.  It separates the mode's rounding noise from implementation error."""
    t = F.conv2d(x, w["stem.conv.weight"], w["stem.conv.bias"], stride=4, padding=2).permute(0, 2, 3, 1)   # NHWC
    t = _ln(t, w["stem.norm.weight"], None, eps)
    for s in range(len(dims)):
        C = dims[s]
        if s > 0:
            u = _ln(t, w["stages.%d.downsample.norm.weight" % s], None, eps)
            u = F.conv2d(u.permute(0, 3, 1, 2), w["stages.%d.downsample.conv.weight" % s], w["stages.%d.downsample.conv.bias" % s],
                         stride=2, padding=1)
            t = u.permute(0, 2, 3, 1)
        B, H, W, _ = t.shape
        q8 = e4m3 and C % 128 == 0
        qa = _q8 if q8 else (lambda a: a)
        qw = _q8w if q8 else (lambda m: m)
        for i in range(depths[s]):
            p = "stages.%d.blocks.%d." % (s, i)
            h = _ln(t, w[p + "norm1.weight"], None, eps)
            if s < attn_from_stage:                                     # SepConv: pw -> StarReLU -> dw 7x7 -> pw
                y = F.linear(h, w[p + "token_mixer.pwconv1.weight"].reshape(2 * C, C))
                y = star_relu(y, w[p + "token_mixer.act1.scale"], w[p + "token_mixer.act1.bias"])
                y = F.conv2d(y.permute(0, 3, 1, 2), w[p + "token_mixer.dwconv.weight"].reshape(2 * C, 1, 7, 7), None, padding=3,
                             groups=2 * C).permute(0, 2, 3, 1)
                y = F.linear(qa(y), qw(w[p + "token_mixer.pwconv2.weight"].reshape(C, 2 * C)))
            else:                                                       # self-attention, head_dim 32, no biases
                heads = C // head_dim
                N = H * W
                qkv = F.linear(h.reshape(B, N, C), w[p + "token_mixer.qkv.weight"]).reshape(B, N, 3, heads, head_dim).permute(2, 0, 3, 1, 4)
                q, k, v = qkv[0], qkv[1], qkv[2]
                a = ((q @ k.transpose(-2, -1)) * (head_dim ** -0.5)).softmax(dim=-1)
                o = (a @ v).transpose(1, 2).reshape(B, N, C)
                y = F.linear(o, w[p + "token_mixer.proj.weight"]).reshape(B, H, W, C)
            rs = w.get(p + "res_scale1.scale")
            t = (t * rs if rs is not None else t) + y
            h = _ln(t, w[p + "norm2.weight"], None, eps)
            y = F.linear(qa(h), qw(w[p + "mlp.fc1.weight"].reshape(4 * C, C)))
            y = star_relu(y, w[p + "mlp.act.scale"], w[p + "mlp.act.bias"])
            y = F.linear(qa(y), qw(w[p + "mlp.fc2.weight"].reshape(C, 4 * C)))
            rs = w.get(p + "res_scale2.scale")
            t = (t * rs if rs is not None else t) + y
    f = t.mean(dim=(1, 2))                                              # global average pool
    return _ln(f, w["head.norm.weight"], w["head.norm.bias"], eps)


def to_torch(weights: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
