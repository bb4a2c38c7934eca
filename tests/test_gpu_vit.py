"""GPU parity: ViT tagger forward (16-bit MFMA operands, fp32 accumulate) vs the float32 torch-CPU oracle.
Tolerance from BASELINE.json's north_star: logits within 1e-3 (absolute, float32).

Default operands are IEEE half since round 3 (ViTTagger); the bf16 mode BASELINE.json names stays covered explicitly
(`operand_f16 = 0`) on the inputs it is good for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3        # north_star: "ViT logits match the ... reference within 1e-3 fp32"


def _oracle_logits(cfg, w, images_u8):
    from oracle import vit as ovit
    x = ovit.preprocess_u8_nhwc(images_u8)
    return ovit.vit_forward(ovit.to_torch(w), x, patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"],
                            gelu_kind="tanh" if cfg["gelu_tanh"] else "erf",
                            pool_then_norm=bool(cfg["pool_then_norm"])).numpy(), x.numpy()


@pytest.mark.parametrize("variant", ["tanh", "erf", "pool_then_norm", "bf16"])
def test_vit_tiny_matches_oracle(variant):
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_TINY)
    if variant == "bf16":
        cfg["operand_f16"] = 0
    if variant == "erf":
        cfg["gelu_tanh"] = 0
    if variant == "pool_then_norm":
        cfg["pool_then_norm"] = 1
    w = synth.vit_weights(cfg, seed=1)
    imgs = synth.images_u8(5, cfg["image_size"], seed=2)
    want, x = _oracle_logits(cfg, w, imgs)
    model = ViTTagger(cfg, w, max_batch=8)
    logits, probs = model.forward_u8(imgs)
    assert np.abs(logits - want).max() <= LOGIT_TOL, np.abs(logits - want).max()
    np.testing.assert_allclose(probs, 1 / (1 + np.exp(-logits.astype(np.float64))), atol=2e-7)
    # float32 NCHW entry point (the tensor tagging.py:174 passes) gives the same result
    logits2, _ = model.forward(x)
    assert np.abs(logits2 - want).max() <= LOGIT_TOL
    # device-resident in/out
    import torch
    dl = torch.empty((5, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    dp = torch.empty_like(dl)
    model.forward_u8(torch.from_numpy(imgs).cuda(), logits=dl, probs=dp)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dl.cpu().numpy(), logits)


@pytest.mark.parametrize("operands", ["half", "bf16"])
def test_vit_b16_448_matches_oracle(operands):
    """config[1] geometry (ViT-B/16 @448, 784 tokens, 10861 classes), 3 noise images (oracle is CPU); both operand types."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_B16_448, operand_f16=1 if operands == "half" else 0)
    w = synth.vit_weights(cfg, seed=0)
    imgs = synth.images_u8(3, 448, seed=1234)
    want, _ = _oracle_logits(cfg, w, imgs)
    model = ViTTagger(cfg, w, max_batch=4)
    logits, _ = model.forward_u8(imgs)
    err = np.abs(logits - want).max()
    print("ViT-B/16@448 %s operands: max |logit error| = %.3e (logit rms %.3f)" % (operands, err, np.sqrt((want ** 2).mean())))
    assert err <= LOGIT_TOL
    assert abs(model.flops_per_image() - 156.78e9) / 156.78e9 < 1e-3


def _errors(got, want):
    """(max |d|, rms d, rms d / rms logit) per image."""
    d = got.astype(np.float64) - want.astype(np.float64)
    rms = np.sqrt((want.astype(np.float64) ** 2).mean(axis=1))
    return np.abs(d).max(axis=1), np.sqrt((d ** 2).mean(axis=1)), np.sqrt((d ** 2).mean(axis=1)) / rms


def test_vit_default_config_structured_images():
    """The bench configuration -- default operands, batch 64, two sub-batch streams -- on STRUCTURED images: one colour, posterised,
    smooth gradient, line art on white, half flat, flat tiles.  Illustrations are made of such regions; every token of a flat region
    carries the same operand rounding error, which the mean pool does not average out (bf16 operands: ~4e-3, asserted below as the
    reason the default changed).  64 images = the six kinds + noise, repeated with different seeds; the oracle (CPU) checks 14."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_B16_448)
    w = synth.vit_weights(cfg, seed=0)
    parts = []
    for rep in range(9):
        parts.append(synth.structured_images_u8(448, seed=100 + rep))
        parts.append(synth.images_u8(1, 448, seed=200 + rep))
    imgs = np.concatenate(parts + [synth.images_u8(1, 448, seed=300)])[:64]
    assert imgs.shape[0] == 64
    check = list(range(14))                                           # two full sets of kinds (+ their noise images)
    want, _ = _oracle_logits(cfg, w, imgs[check])
    model = ViTTagger(cfg, w, max_batch=64)                            # default operands (IEEE half), two streams at batch 64
    logits, _ = model.forward_u8(imgs)
    mx, rms, rel = _errors(logits[check], want)
    kinds = list(synth.STRUCTURED_KINDS) + ["noise"]
    for i in check:
        print("default operands, %-10s max |dlogit| %.3e  rms %.3e  rms-relative %.3e" % (kinds[i % 7], mx[i], rms[i], rel[i]))
    assert mx.max() <= LOGIT_TOL, mx
    again, _ = model.forward_u8(imgs)
    np.testing.assert_array_equal(again, logits)
    model.close()
    # the opt-in bf16 mode on the same images: inside the tolerance on noise, outside it on flat regions -- why it is not the default
    cfg0 = dict(cfg, operand_f16=0)
    m0 = ViTTagger(cfg0, w, max_batch=64)
    l0, _ = m0.forward_u8(imgs)
    mx0, _, _ = _errors(l0[check], want)
    for i in check[:7]:
        print("bf16 operands,    %-10s max |dlogit| %.3e" % (kinds[i % 7], mx0[i]))
    assert mx0[6] <= LOGIT_TOL and mx0[13] <= LOGIT_TOL               # noise
    assert mx0.max() <= 2e-2                                           # bounded, but ...
    assert mx0[0] > mx[0]                                              # ... the flat image is where half operands matter


def test_vit_trained_like_checkpoint():
    """A checkpoint in the regime a trained tagger runs in (synth.vit_weights(trained_like=True)): peaked attention with a heavy
    tail (log2-domain scores far above 16, where an IEEE-half 2^S is +inf: the fixed-reference softmax and, for the heavy head, its
    classic fallback run), logit rms ~10, sparse probabilities, tens of labels selected.  The absolute 1e-3 of BASELINE.json is
    stated for logits; at a logit scale 30x the random init's the same RELATIVE accuracy is 30x the absolute error, so this test
    reports both and asserts the relative one plus the labels actually selected."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import TagSelector, ViTTagger
    from oracle import tags as otags
    cfg = dict(synth.VIT_B16_448)
    w = synth.vit_weights(cfg, seed=0, trained_like=True)
    imgs = np.concatenate([synth.images_u8(2, 448, seed=5), synth.structured_images_u8(448, seed=77)])
    want, _ = _oracle_logits(cfg, w, imgs)
    model = ViTTagger(cfg, w, max_batch=8)
    logits, probs = model.forward_u8(imgs)
    assert np.isfinite(logits).all()
    mx, rms, rel = _errors(logits, want)
    kinds = ["noise", "noise"] + list(synth.STRUCTURED_KINDS)
    print("trained-like checkpoint: logit rms %.2f" % np.sqrt((want ** 2).mean()))
    for i in range(len(imgs)):
        print("  %-10s max |dlogit| %.3e  rms %.3e  rms-relative %.3e" % (kinds[i], mx[i], rms[i], rel[i]))
    # measured (round 4, profiles/r04_precision_vit.json): rms-relative <= 6.7e-5, max |dlogit| 5.2e-4 .. 3.0e-3 (half flat; flat 1.7e-3) --
    # north_star's absolute 1e-3 is NOT met on four of the eight kinds at this logit scale; DESIGN.md section 2 has the attribution
    # per rounded operand and what meeting it costs (operand_f16 bit 4 removes the flat-image class for 5 %; everything else > 40 %)
    assert rel.max() <= 1.5e-4, rel
    assert mx.max() <= 5e-3, mx
    assert np.median(mx) <= 1.2e-3, mx
    # what the product outputs: the selected labels (MCut on both categories, tagging.py:333) equal the oracle's on every image
    names, cat = synth.label_table(cfg["num_classes"])
    sel = TagSelector(cat, max_batch=8)
    counts, ids, _ = sel.run(probs, 0.3, True, 0.3, True)
    want_probs = otags.sigmoid_f32(want)
    gi, ci = list(np.where(cat == 0)[0]), list(np.where(cat == 4)[0])
    n_sel = []
    for i in range(len(imgs)):
        g, c, _, _ = otags.select_indices(want_probs[i], gi, ci, 0.3, True, 0.3, True)
        got = list(ids[i, :counts[i, 0] + counts[i, 1]])
        n_sel.append(len(got))
        assert sorted(got) == sorted(list(g) + list(c)), (i, got, list(g), list(c))
    print("  labels selected per image:", n_sel)
    assert 10 <= min(n_sel) and max(n_sel) <= 60


def test_vit_split_attention_output():
    """operand_f16 = 1 | 16 (HIPTS_OPERAND_SPLIT_ATT): the attention output reaches the output projection as a hi | lo pair.  What it is
    for: an image of ONE colour -- every token's attention output then carries the same rounding, the error class that dominates the
    flat image at a trained tagger's logit scale (tools/logit_attribution.py: 1.8e-3 of the 1.75e-3).  Asserted on the trained-like
    checkpoint: the flat image's error falls below 5e-4 (measured 2.2e-4, from 1.7e-3), no image kind gets worse by more than the
    run-to-run spread of this statistic, and the labels selected are the oracle's.  The fallback softmax writes both halves as well
    (the checkpoint's heavy heads take it)."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_B16_448)
    w = synth.vit_weights(cfg, seed=0, trained_like=True)
    imgs = np.concatenate([synth.images_u8(2, 448, seed=5), synth.structured_images_u8(448, seed=77)])
    kinds = ["noise", "noise"] + list(synth.STRUCTURED_KINDS)
    want, _ = _oracle_logits(cfg, w, imgs)
    base = ViTTagger(cfg, w, max_batch=8)
    l1, _ = base.forward_u8(imgs)
    base.close()
    model = ViTTagger(dict(cfg, operand_f16=1 | 16), w, max_batch=8)
    l17, _ = model.forward_u8(imgs)
    again, _ = model.forward_u8(imgs)
    np.testing.assert_array_equal(l17, again)
    model.close()
    m1, _, _ = _errors(l1, want)
    m17, _, _ = _errors(l17, want)
    for i, k in enumerate(kinds):
        print("  %-10s max |dlogit| half %.3e  half + split attention output %.3e" % (k, m1[i], m17[i]))
    flat = kinds.index("flat")
    assert m17[flat] <= 5e-4 and m17[flat] < 0.4 * m1[flat], (m1[flat], m17[flat])
    assert (m17 <= np.maximum(1.5 * m1, 1e-3)).all(), (m1, m17)


def test_vit_requires_all_tensors():
    import hiptagsearch
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_TINY)
    w = synth.vit_weights(cfg, seed=1)
    del w["blocks.1.mlp.fc2.bias"]
    model = ViTTagger(cfg, w, max_batch=2)
    with pytest.raises(hiptagsearch.HipTagSearchError):
        model.forward_u8(synth.images_u8(1, cfg["image_size"]))


def test_deferred_join_gives_the_same_outputs():
    """hipts_vit_set_deferred_join: forward() does not make the caller's stream wait for the sub-batch streams;
    hipts_vit_join does, on whichever stream consumes the outputs.  Two forwards back to back, joined on a
    side stream, must give the batch's ordinary outputs."""
    import torch
    from hiptagsearch import _lib, synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_TINY)
    w = synth.vit_weights(cfg, seed=1)
    model = ViTTagger(cfg, w, max_batch=32)
    imgs = [torch.from_numpy(synth.images_u8(32, cfg["image_size"], seed=s)).cuda() for s in (21, 22)]     # 32 images -> two sub-batches
    want = []
    for im in imgs:
        p = torch.empty((32, cfg["num_classes"]), dtype=torch.float32, device="cuda")
        model.forward_u8(im, probs=p, want="probs")
        torch.cuda.synchronize()
        want.append(p.cpu().numpy())
    _lib.call("hipts_vit_set_deferred_join", model._h, 1)
    side = torch.cuda.Stream()
    outs = [torch.empty((32, cfg["num_classes"]), dtype=torch.float32, device="cuda") for _ in imgs]
    copies = []
    for im, o in zip(imgs, outs):
        model.forward_u8(im, probs=o, want="probs")
        with torch.cuda.stream(side):
            _lib.call("hipts_vit_join", model._h, _lib.current_stream_ptr())
            copies.append(o.clone())                       # consumer on the side stream
    torch.cuda.synchronize()
    _lib.call("hipts_vit_set_deferred_join", model._h, 0)
    for c, wnt in zip(copies, want):
        np.testing.assert_array_equal(c.cpu().numpy(), wnt)


def test_folded_layernorm_path_matches_the_separate_kernels():
    """HIPTS_LN_FOLD=1 (read when the handle is created, hence the child interpreter): LayerNorms prepared by the residual
    GEMM epilogues and applied in the consumer epilogues.  Same tolerance against the oracle as the default path, run to run
    identical, and close to the default path's logits (both round the same operands, in a different place)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from hiptagsearch import synth
from hiptagsearch.tagger import ViTTagger
from oracle import vit as ov
cfg = dict(synth.VIT_B16_448); cfg["depth"] = 4
w = synth.vit_weights(cfg, seed=7)
imgs = synth.images_u8(20, 448, seed=8)            # 20 images: two sub-batch streams of 10
m = ViTTagger(cfg, w, max_batch=32)
got, _ = m.forward_u8(imgs)
again, _ = m.forward_u8(imgs)
assert np.array_equal(got, again)
one, _ = m.forward_u8(imgs[3:4])
assert np.array_equal(one[0], got[3])
want = ov.vit_forward(ov.to_torch(w), ov.preprocess_u8_nhwc(imgs[:2]), patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"]).numpy()
err = np.abs(got[:2] - want).max()
print("FOLD", float(err))
np.save(sys.argv[1], got)
assert err <= 1e-3, err
""" % (root, os.path.join(root, "anime-illust-image-searcher_amd"))
    import tempfile
    outs = []
    for fold in ("1", "0"):
        f = tempfile.mktemp(suffix=".npy")
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, env=dict(os.environ, HIPTS_LN_FOLD=fold), timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
        outs.append(np.load(f))
        os.remove(f)
    assert not np.array_equal(outs[0], outs[1])               # the folded path really ran
    assert np.abs(outs[0] - outs[1]).max() <= 1e-3
