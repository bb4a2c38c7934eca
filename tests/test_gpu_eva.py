"""GPU parity: EVA02 tagger forward (patch 14, class token, 2-D RoPE, SwiGLU with inner LayerNorm) vs the float32
torch-CPU oracle (oracle/eva.py; parity unpinned: timm and the weights are not available here).

Tolerance: the north_star's 1e-3 on logits is stated for the ViT-B/16 contract model.  For this model the
synthetic checkpoint (unit-variance LayerNorm outputs after every SwiGLU, 24 blocks of width 1024) makes the
residual branches O(1), so 16-bit operand rounding reaches the logits at full size.  Measured (round 2, LayerNorms and RoPE
folded into the GEMM epilogues): IEEE-half operands -- the DEFAULT of EvaTagger -- max |dlogit| 2.9e-4 (tiny) / 2.1e-3
(EVA02-L, logit rms 0.65); bf16 operands 2.3e-3 / 1.7e-2: the 8x ratio of the two mantissas, i.e. rounding only.
Bounds = measured + 20 %: half 5e-4 / 3e-3 (2.1e-3 on the default schedule, 2.6e-3 with HIPTS_EVA_ROWSTAT_KERNEL=1 and one stream: the
order in which the LayerNorm partial sums are added), bf16 4e-3 / 2.1e-2, and cosine(logits, oracle) >= 0.9995 (bf16) / 0.99999 (half)."""

TOL = {"tiny": {0: 4e-3, 1: 5e-4}, "large": {0: 2.1e-2, 1: 3e-3}}
COS = {0: 0.9995, 1: 0.99999}


def _cos(a, b):
    return ((a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))).min()
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(cfg, w, images_u8):
    from oracle import eva as oe, vit as ovit
    x = ovit.preprocess_u8_nhwc(images_u8)
    return oe.eva_forward(oe.to_torch(w), x, patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"], ref_grid=cfg["rope_ref_grid"]).numpy(), x.numpy()


@pytest.mark.parametrize("f16", [0, 1])
def test_eva_tiny_matches_oracle(f16):
    from hiptagsearch import synth
    from hiptagsearch.tagger import EvaTagger
    cfg = dict(synth.EVA02_TINY, operand_f16=f16)
    w = synth.eva_weights(cfg, seed=1)
    imgs = synth.images_u8(5, cfg["image_size"], seed=2)
    want, x = _oracle(cfg, w, imgs)
    model = EvaTagger(cfg, w, max_batch=8)
    logits, probs = model.forward_u8(imgs)
    err = np.abs(logits - want).max()
    print("EVA tiny (operand_f16=%d) max |logit error| = %.3e" % (f16, err))
    assert err <= TOL["tiny"][f16], err
    assert _cos(logits, want) >= COS[f16]
    np.testing.assert_allclose(probs, 1 / (1 + np.exp(-logits.astype(np.float64))), atol=2e-7)
    logits2, _ = model.forward(x)                                  # float32 NCHW entry point (tagging.py:174's tensor)
    assert np.abs(logits2 - want).max() <= TOL["tiny"][f16]
    np.testing.assert_array_equal(model.forward_u8(imgs)[0], logits)   # run to run identical
    # a batch computed image by image gives the same rows
    one = np.concatenate([model.forward_u8(imgs[i:i + 1])[0] for i in range(2)])
    np.testing.assert_array_equal(one, logits[:2])


@pytest.mark.parametrize("f16", [0, 1])
def test_eva02_large_448_matches_oracle(f16):
    """The reference's model geometry: EVA02-L/14 @448 (1025 tokens, width 1024, 24 blocks, hidden 2730), 2 images."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import EvaTagger
    cfg = dict(synth.EVA02_L14_448, operand_f16=f16)
    w = synth.eva_weights(cfg, seed=0)
    imgs = synth.images_u8(2, 448, seed=1234)
    want, _ = _oracle(cfg, w, imgs)
    model = EvaTagger(cfg, w, max_batch=2)
    logits, _ = model.forward_u8(imgs)
    err = np.abs(logits - want).max()
    print("EVA02-L/14@448 max |logit error| = %.3e (logit rms %.3f), %.1f GFLOP/image" % (err, np.sqrt((want ** 2).mean()), model.flops_per_image() / 1e9))
    assert err <= TOL["large"][f16], err
    assert _cos(logits, want) >= COS[f16]
    if f16 == 1:                                                   # half operands are what EvaTagger runs unless told otherwise
        dflt = EvaTagger(dict(synth.EVA02_L14_448), w, max_batch=2)
        np.testing.assert_array_equal(dflt.forward_u8(imgs)[0], logits)


def test_eva02_large_trained_like_checkpoint():
    """EVA02-L/14 (the model the reference loads) in the regime a trained tagger runs in -- synth.eva_weights(trained_like=True): peaked
    attention with a heavy tail (the half-safe softmax and its fallback run), logit rms ~10, tens of labels selected -- on a noise image, a
    flat one and a cel-shaded one.  Absolute AND rms-relative logit error are reported; the relative one and the selected labels are
    asserted (the absolute 1e-3 of BASELINE.json is stated at the random init's logit scale, 15x smaller)."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import EvaTagger, TagSelector
    from oracle import tags as otags
    cfg = dict(synth.EVA02_L14_448)
    w = synth.eva_weights(cfg, seed=0, trained_like=True)
    imgs = np.concatenate([synth.images_u8(1, 448, seed=5), synth.structured_images_u8(448, seed=77, kinds=("flat", "blocks"))])
    want, _ = _oracle(cfg, w, imgs)
    model = EvaTagger(cfg, w, max_batch=4)
    logits, probs = model.forward_u8(imgs)
    assert np.isfinite(logits).all()
    d = logits.astype(np.float64) - want.astype(np.float64)
    rms_l = np.sqrt((want.astype(np.float64) ** 2).mean(axis=1))
    mx, rel = np.abs(d).max(axis=1), np.sqrt((d ** 2).mean(axis=1)) / rms_l
    for name, a, r in zip(("noise", "flat", "blocks"), mx, rel):
        print("EVA02-L trained-like, %-6s max |dlogit| %.3e  rms-relative %.3e  (logit rms %.2f)" % (name, a, r, rms_l.mean()))
    # Bounds = what is measured (noise 2.6e-2 / rms-relative 6.1e-4, flat 2.5e-3, blocks 9e-3; bench.py's eva02_large.oracle_check carries the
    # same numbers) + 25 %.  Attribution (tools/logit_attribution.py --model eva, profiles/r05_eva_attribution.txt): every 16-bit operand carries
    # about 1e-2 of it on the noise image (xn1 1.3e-2, q|k|v 1.5e-2, xn2 1.1e-2, hmid 6.6e-3, att 4.5e-3, P 1.9e-3) -- amplification through
    # 24 blocks of peaked attention, not one operand: the cheapest single hi | lo split (q, k, v) would buy 19 %.
    assert rel.max() <= 8e-4 and mx.max() <= 3.3e-2, (mx, rel)
    names, cat = synth.label_table(cfg["num_classes"])
    counts, ids, _ = TagSelector(cat, max_batch=4).run(probs, 0.3, True, 0.3, True)
    gi, ci = list(np.where(cat == 0)[0]), list(np.where(cat == 4)[0])
    wp = otags.sigmoid_f32(want)
    for i in range(len(imgs)):
        g, c, _, _ = otags.select_indices(wp[i], gi, ci, 0.3, True, 0.3, True)
        got = sorted(ids[i, :counts[i, 0] + counts[i, 1]])
        assert got == sorted(list(g) + list(c)), i
        assert 10 <= len(got) <= 60


def test_eva_missing_tensor_is_reported():
    from hiptagsearch import synth, _lib
    from hiptagsearch.tagger import EvaTagger
    cfg = dict(synth.EVA02_TINY)
    w = synth.eva_weights(cfg, seed=1)
    w.pop("blocks.1.attn.k_proj.weight")
    with pytest.raises(_lib.HipTagSearchError, match="not set"):
        EvaTagger(cfg, w, max_batch=2).forward_u8(synth.images_u8(1, cfg["image_size"], seed=3))


def test_eva_two_sub_batch_streams_match_single_stream():
    """batch >= 16 runs as two sub-batches on two internal streams (per-image workspace offsets, folded-LayerNorm statistics
    per sub-batch): every image must come out exactly as when it is computed alone."""
    from hiptagsearch import synth
    from hiptagsearch.tagger import EvaTagger
    cfg = dict(synth.EVA02_TINY)
    w = synth.eva_weights(cfg, seed=4)
    imgs = synth.images_u8(19, cfg["image_size"], seed=5)
    model = EvaTagger(cfg, w, max_batch=32)
    logits, _ = model.forward_u8(imgs)
    np.testing.assert_array_equal(model.forward_u8(imgs)[0], logits)
    for i in (0, 9, 10, 18):                                      # both halves, including their first and last images
        np.testing.assert_array_equal(model.forward_u8(imgs[i:i + 1])[0][0], logits[i])
