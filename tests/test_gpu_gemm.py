"""GPU: the bf16 MFMA GEMM main loops (ping-pong 16/32-MFMA segments, three-stage, two-barrier) against
a float64 product of the same bf16 operands, on ragged and exact shapes, plus run-to-run determinism of
the whole forward (a race in the LDS-DMA pipeline shows up as nondeterminism before it shows as error)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [(256, 256, 64), (1, 16, 64), (300, 272, 128), (784, 768, 768), (1000, 2304, 768), (513, 3072, 192), (777, 768, 3072)]

_CHILD = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, %(pkg)r)
from hiptagsearch import _lib, synth
lib = _lib.load()
f = lib.hiptsdbg_gemm_run
f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
bad = []
for (M, N, K) in %(shapes)r:
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = synth.round_to_bf16(rng.standard_normal((M, K)).astype(np.float32))
    w = synth.round_to_bf16((rng.standard_normal((N, K)) * 0.05).astype(np.float32))
    a16 = (a.view(np.uint32) >> 16).astype(np.uint16); w16 = (w.view(np.uint32) >> 16).astype(np.uint16)
    out = np.empty((M, N), np.float32); out2 = np.empty((M, N), np.float32)
    for o in (out, out2):
        st = f(M, N, K, a16.ctypes.data, w16.ctypes.data, o.ctypes.data)
        assert st == 0, _lib.last_error()
    want = a.astype(np.float64) @ w.astype(np.float64).T
    err = np.abs(out - want).max()
    tol = 2e-6 * K ** 0.5 * 4 + 1e-6          # fp32 accumulation of exact bf16 products
    if err > tol or not np.array_equal(out, out2):
        bad.append((M, N, K, float(err), tol, bool(np.array_equal(out, out2))))
print("BAD", bad)
sys.exit(1 if bad else 0)
"""


@pytest.mark.parametrize("variant", ["pp", "dw", "pp2", "s3", "v1"])
def test_gemm_variants_match_float64(variant):
    """Each variant is selected per process (HIPTS_GEMM is read once), hence the child interpreter."""
    env = dict(os.environ, HIPTS_GEMM=variant)
    code = _CHILD % {"pkg": os.path.join(ROOT, "anime-illust-image-searcher_amd"), "shapes": SHAPES}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_gemm_default_dispatch_persistent_and_underfilled():
    """No HIPTS_GEMM: the dispatch the product uses.  (70000, 768, 128) is 822 tiles on 256 persistent
    workgroups (several tiles per workgroup, next-tile prefetch); (11520, 512, 512) has 90 tiles < CUs and
    goes to the two-per-CU kernel; (300, 272, 128) stays a plain one-tile-per-workgroup launch."""
    env = {k: v for k, v in os.environ.items() if k != "HIPTS_GEMM"}
    code = _CHILD % {"pkg": os.path.join(ROOT, "anime-illust-image-searcher_amd"),
                     "shapes": [(70000, 768, 128), (11520, 512, 512), (300, 272, 128), (50000, 208, 64),
                                # odd and even K-tile counts on persistent grids (the two-K-tiles-ahead staging toggles its LDS stage per tile)
                                (66000, 768, 192), (66000, 1024, 320), (40000, 768, 1536)]}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_gemm_half_operands_and_tile_heights():
    """IEEE-half operands (the product default) through the persistent loop at each tile height the launcher picks:
    (10250, 1024, 1024) = EVA02-L's proj at the reference batch of 10 -> 192-row tiles (216 tiles, one round);
    (50176, 768, 768) -> 224 rows; (66000, 1024, 320) -> 256 rows; (1025, 1024, 1024) one ragged round of 192."""
    env = {k: v for k, v in os.environ.items() if k != "HIPTS_GEMM"}
    env["HIPTS_DBG_GEMM_F16"] = "1"
    code = _CHILD % {"pkg": os.path.join(ROOT, "anime-illust-image-searcher_amd"),
                     "shapes": [(10250, 1024, 1024), (50176, 768, 768), (66000, 1024, 320), (1025, 1024, 1024), (10250, 1024, 2752)]}
    code = code.replace("a = synth.round_to_bf16(rng.standard_normal((M, K)).astype(np.float32))",
                        "a = rng.standard_normal((M, K)).astype(np.float16).astype(np.float32)")
    code = code.replace("w = synth.round_to_bf16((rng.standard_normal((N, K)) * 0.05).astype(np.float32))",
                        "w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16).astype(np.float32)")
    code = code.replace("a16 = (a.view(np.uint32) >> 16).astype(np.uint16); w16 = (w.view(np.uint32) >> 16).astype(np.uint16)",
                        "a16 = a.astype(np.float16).view(np.uint16); w16 = w.astype(np.float16).view(np.uint16)")
    assert "np.float16" in code and "round_to_bf16" not in code
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("half", [0, 1])
def test_gemm_split_k_tail(half):
    """The split-K tail of the residual GEMMs (GemmArgs::sk_*): work items past the last full round are (tile, K slice) pairs, slices leave
    their accumulators in slabs, the last arriver adds them in slice order and runs the epilogue.  Against float64, twice (bit-equal: the
    sum does not depend on who arrives last), two launches per run on one workspace (the tickets return to zero):
    (5125, 1024, 1024) = EVA02-L's proj per sub-batch of 5 images, 84 tiles x 3 slices; (5125, 1024, 2752) its fc2; (25088, 768, 3072) =
    the ViT's fc2 per 32 images, 256 whole tiles + 38 x 4 slices; (777, 768, 3072) ragged rows, 12 tiles x 4; (300, 272, 128) too short
    to split (one launch path must survive the workspace being there)."""
    env = {k: v for k, v in os.environ.items() if k != "HIPTS_GEMM"}
    env["HIPTS_DBG_GEMM_SK"] = "1"
    env["HIPTS_GEMM_SPLITK"] = "4"              # opt-in: the split is off by default (it gives up batch invariance, gemm.hip launcher)
    shapes = [(5125, 1024, 1024), (5125, 1024, 2752), (25088, 768, 3072), (777, 768, 3072), (300, 272, 128), (25088, 768, 768)]
    code = _CHILD % {"pkg": os.path.join(ROOT, "anime-illust-image-searcher_amd"), "shapes": shapes}
    # the runner launches twice into the same zero-initialised output: out = 2 A W^T
    code = code.replace("want = a.astype(np.float64) @ w.astype(np.float64).T", "want = 2.0 * (a.astype(np.float64) @ w.astype(np.float64).T)")
    code = code.replace("tol = 2e-6 * K ** 0.5 * 4 + 1e-6", "tol = 2 * (2e-6 * K ** 0.5 * 4 + 1e-6)")
    if half:
        env["HIPTS_DBG_GEMM_F16"] = "1"
        code = code.replace("a = synth.round_to_bf16(rng.standard_normal((M, K)).astype(np.float32))",
                            "a = rng.standard_normal((M, K)).astype(np.float16).astype(np.float32)")
        code = code.replace("w = synth.round_to_bf16((rng.standard_normal((N, K)) * 0.05).astype(np.float32))",
                            "w = (rng.standard_normal((N, K)) * 0.05).astype(np.float16).astype(np.float32)")
        code = code.replace("a16 = (a.view(np.uint32) >> 16).astype(np.uint16); w16 = (w.view(np.uint32) >> 16).astype(np.uint16)",
                            "a16 = a.astype(np.float16).view(np.uint16); w16 = w.astype(np.float16).view(np.uint16)")
        assert "round_to_bf16" not in code
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_forward_is_deterministic_and_batch_invariant():
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_B16_448)
    cfg["depth"] = 3                                       # full-width layers, short enough to repeat
    w = synth.vit_weights(cfg, seed=5)
    imgs = synth.images_u8(9, 448, seed=6)
    model = ViTTagger(cfg, w, max_batch=16)
    ref, _ = model.forward_u8(imgs)
    for _ in range(4):
        again, _ = model.forward_u8(imgs)
        np.testing.assert_array_equal(again, ref)          # bitwise: no race, no atomics in the forward
    one, _ = model.forward_u8(imgs[4:5])                   # an image's logits do not depend on its batch
    np.testing.assert_array_equal(one[0], ref[4])


def test_gemm_e4m3_operands_match_float64():
    """configs[4] names fp8 MFMA: the e4m3 operand path (v_mfma_scale_f32_16x16x128_f8f6f4) against a float64 product
    of the same quantised operands (torch's float8_e4m3fn conversion is the independent quantiser), fp32 and e4m3 outputs."""
    import torch
    from hiptagsearch import _lib
    lib = _lib.load()
    f = lib.hiptsdbg_gemm8_run
    f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for (M, N, K) in [(256, 256, 128), (1, 16, 128), (300, 272, 256), (3000, 768, 768), (777, 512, 2048), (70000, 256, 128)]:
        rng = np.random.default_rng(M + 3 * N + 7 * K)
        a = rng.standard_normal((M, K)).astype(np.float32)
        w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
        e = ctypes.c_int(0)
        out = np.empty((M, N), np.float32)
        assert f(M, N, K, a.ctypes.data, w.ctypes.data, 0, out.ctypes.data, ctypes.byref(e)) == 0, _lib.last_error()
        assert 224.0 < np.abs(w).max() * 2.0 ** e.value <= 448.0
        aq = torch.from_numpy(a).to(torch.float8_e4m3fn).to(torch.float64).numpy()
        wq = torch.from_numpy(w * np.float32(2.0 ** e.value)).to(torch.float8_e4m3fn).to(torch.float64).numpy() * 2.0 ** -e.value
        want = aq @ wq.T
        err = np.abs(out - want).max()
        assert err <= 2e-6 * K ** 0.5 * 4 * np.abs(want).max() + 1e-6, (M, N, K, err)
        out8 = np.empty((M, N), np.uint8)
        assert f(M, N, K, a.ctypes.data, w.ctypes.data, 1, out8.ctypes.data, ctypes.byref(e)) == 0, _lib.last_error()
        got8 = torch.from_numpy(out8).view(torch.float8_e4m3fn).to(torch.float32).numpy()
        want8 = torch.from_numpy(out).to(torch.float8_e4m3fn).to(torch.float32).numpy()      # RNE of the fp32 result
        np.testing.assert_array_equal(got8, want8)


@pytest.mark.parametrize("form", ["erf", "tanh"])
def test_fc1_epilogue_gelu_against_float64(form):
    """The GELU the fc1 epilogue applies (gemm.hip gelu_f4): the erf form is a rational approximation in packed fp32 (round 4; libm's
    erff before), the tanh form one exp2 and one reciprocal per value.  Against the float64 definitions (timm's nn.GELU / GELU(tanh),
    tagging.py:174's model.forward) on a dense grid, the tails and the clamp point: within 3e-7 max(1, |x|) -- the fp32 expression
    0.5 x (1 + erf(x / sqrt 2)) itself is within 1.1e-7 -- where what is stored is a 16-bit value (4.9e-4 relative)."""
    import ctypes
    from math import sqrt, pi
    from scipy.special import erf
    sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
    from hiptagsearch import _lib
    lib = _lib.load()
    f = lib.hiptsdbg_gelu
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(3)
    x = np.concatenate([np.linspace(-12, 12, 400001), rng.standard_normal(200000) * 2.5, [0.0, -0.0, 5.5, -5.5, 5.4999995, 40.0, -40.0, 1e-30, -1e-30, 6e4, -6e4]])
    x = np.resize(x.astype(np.float32), (len(x) + 3) // 4 * 4)
    y = np.empty_like(x)
    st = f(x.ctypes.data, len(x), 1 if form == "tanh" else 0, y.ctypes.data)
    assert st == 0, _lib.last_error()
    xd = x.astype(np.float64)
    want = 0.5 * xd * (1 + np.tanh(sqrt(2 / pi) * (xd + 0.044715 * xd ** 3))) if form == "tanh" else 0.5 * xd * (1 + erf(xd / sqrt(2)))
    err = np.abs(y - want) / np.maximum(1.0, np.abs(xd))
    assert np.isfinite(y).all()
    assert err.max() <= 3e-7, (err.max(), x[err.argmax()])
    assert (y[x >= 0] >= 0).all() and (np.abs(y[x > 6] / x[x > 6] - 1) <= 1.2e-7).all()


def test_four_wave_loop_equals_eight_wave_loop_bit_for_bit():
    """csrc/gemm4.hip (4 waves, one per SIMD, 128 x 128 per wave, accumulators in AGPRs, inline-asm MFMA / LDS / LDS-DMA stream; round 5, off by
    default: HIPTS_GEMM_Q4) hands its accumulators to the same epilogues as gemm_pp_kernel and runs an element's MFMA chain over K in the
    same order: every output buffer (16-bit GELU / q | k | v tensors, the fp32 residual stream, its 16-bit gamma * x copy, the row sums) must
    agree byte for byte with the 8-wave loop on the same operands -- persistent grids (several tiles per workgroup: the K-tile stream runs
    across tiles), 2 .. 48 K-tiles, both operand types."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
    from hiptagsearch import _lib
    lib = _lib.load()
    f = lib.hiptsdbg_gemm_q4_compare
    f.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_float)]
    cases = [(4, 25088, 3072, 128), (4, 25088, 1024, 768), (1, 25088, 2304, 256), (1, 25088, 1536, 128), (13, 25088, 768, 768), (13, 25088, 768, 3072)]
    for f16 in (1, 0):
        for epi, M, N, K in cases:
            bad = ctypes.c_longlong(-1)
            ms = (ctypes.c_float * 2)()
            st = f(M, N, K, epi, f16, 1, ctypes.byref(bad), ms)
            assert st == 0, (epi, M, N, K, f16, _lib.last_error())
            assert bad.value == 0, (epi, M, N, K, f16, bad.value)
