"""The ViT oracle against an implementation by somebody else (CPU, no GPU).

oracle/vit.py is a restatement of timm's VisionTransformer (timm is not installed; the reference holds no fixtures for the
forward).  HuggingFace transformers IS in this image, and its ViT is the conversion target of timm's ViT checkpoints: the same
pre-norm block, written independently.  Loading the oracle's synthetic weights into `transformers` ViTLayer modules (fused qkv
split into q / k / v) and running the same embedded tokens through both must agree to float32 rounding -- block wiring, attention
scale, softmax axis, GELU form, LayerNorm epsilon.  What this does NOT pin: the timm-specific head of the tagger (no class token,
final norm then average pool) and the preprocessing, which stay restatements anchored on the reference's call sites."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))

transformers = pytest.importorskip("transformers")


def _hf_layers(cfg, w):
    from transformers import ViTConfig, ViTModel
    D, depth = cfg["dim"], cfg["depth"]
    hf = ViTConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=cfg["heads"], intermediate_size=cfg["mlp_dim"],
                   hidden_act="gelu_pytorch_tanh" if cfg.get("gelu_tanh", 1) else "gelu", layer_norm_eps=cfg["ln_eps"], qkv_bias=True,
                   hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, image_size=cfg["image_size"], patch_size=cfg["patch"])
    model = ViTModel(hf, add_pooling_layer=False).eval()
    sd = {}
    for i in range(depth):
        p, q = "blocks.%d." % i, "layers.%d." % i
        qkv_w, qkv_b = w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):          # timm's fused rows are [q; k; v]
            sd[q + "attention.%s.weight" % n] = qkv_w[j * D:(j + 1) * D]
            sd[q + "attention.%s.bias" % n] = qkv_b[j * D:(j + 1) * D]
        sd[q + "attention.o_proj.weight"] = w[p + "attn.proj.weight"]
        sd[q + "attention.o_proj.bias"] = w[p + "attn.proj.bias"]
        sd[q + "layernorm_before.weight"], sd[q + "layernorm_before.bias"] = w[p + "norm1.weight"], w[p + "norm1.bias"]
        sd[q + "layernorm_after.weight"], sd[q + "layernorm_after.bias"] = w[p + "norm2.weight"], w[p + "norm2.bias"]
        sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"] = w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]
        sd[q + "mlp.fc2.weight"], sd[q + "mlp.fc2.bias"] = w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(not m.startswith("layers.") for m in missing), missing      # every block parameter came from the oracle's weights
    return model


@pytest.mark.parametrize("gelu_tanh", [1, 0])
def test_vit_blocks_agree_with_transformers(gelu_tanh):
    from hiptagsearch import synth
    from oracle import vit as ovit
    cfg = dict(synth.VIT_TINY, depth=3, gelu_tanh=gelu_tanh)
    w = ovit.to_torch(synth.vit_weights(cfg, seed=3))
    rng = np.random.default_rng(0)
    images = rng.integers(0, 256, (2, cfg["image_size"], cfg["image_size"], 3), dtype=np.uint8)
    x = ovit.preprocess_u8_nhwc(images)
    want = ovit.vit_forward(w, x, patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"], gelu_kind="tanh" if gelu_tanh else "erf",
                            return_tokens=True)
    # the same embedded tokens through transformers' blocks
    t = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=cfg["patch"]).flatten(2).transpose(1, 2) + w["pos_embed"]
    model = _hf_layers(cfg, w)
    with torch.no_grad():
        for layer in model.layers:
            out = layer(t)
            t = out[0] if isinstance(out, tuple) else out
    err = (t - want).abs().max().item()
    scale = want.abs().max().item()
    print("oracle vs transformers %s ViT blocks: max |diff| %.3e on values up to %.2f" % (transformers.__version__, err, scale))
    assert err <= 2e-5 * max(scale, 1.0)
    # and the tagger's tail on top of either: final norm on the tokens, average pool, head (timm fc_norm=False) -- identical inputs
    logits = ovit.vit_forward(w, x, patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"], gelu_kind="tanh" if gelu_tanh else "erf")
    tail = F.linear(F.layer_norm(t, (cfg["dim"],), w["norm.weight"], w["norm.bias"], cfg["ln_eps"]).mean(dim=1), w["head.weight"], w["head.bias"])
    assert (tail - logits).abs().max().item() <= 1e-4
