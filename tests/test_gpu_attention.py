"""GPU: the attention kernels alone against a float64 softmax -- including inputs that FORCE the fallback of the max-free fast path
(cdna_hip_programming.md rule 26: a rare data-dependent branch needs its own test).

Two kernels: csrc/attn2.hip (head_dim 64: the ViT and EVA02 forwards since round 3; V in its natural layout; hiptsdbg_attention2, geometry
variants 5 = default, 3, 1) and csrc/attn.hip (head_dim 32 and 64, V transposed: the CAFormer's; hiptsdbg_attention_run).

Fast path: no running maximum.  bf16 operands: P = 2^S, valid while the row sum stays inside [2^-100, 2^100]; IEEE-half operands:
P = 2^(S - m_ref) with m_ref the row's maximum over the first key tile, valid while the row sum stays below 2^15.  Otherwise the
workgroup repeats the block with the classic per-tile maximum (attn.hip)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bf16(x):
    from hiptagsearch import synth
    return synth.round_to_bf16(np.asarray(x, dtype=np.float32))


def _op(x, f16):
    """float32 values representable in the 16-bit operand type (bf16, or IEEE half with f16)."""
    if f16:
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    return _bf16(x)


def _bits(x, f16=0):
    if f16:
        return np.ascontiguousarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _from_bits(b, f16=0):
    if f16:
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


KERNELS64 = ["attn2", "attn2:6", "attn2:3", "attn2:1", "attn"]      # head_dim 64: the round-3 kernel (default geometry and two others) and the round-2 one


def _run(q, k, v, tokens, hd, f16=0, kernel="attn"):
    """q, k, v: float32 (representable in the operand type) [BH, tokens, hd]; q already in the log2 domain.  Returns float32 [BH, tokens, hd]."""
    from hiptagsearch import _lib
    lib = _lib.load()
    BH = q.shape[0]
    tp = (tokens + 63) // 64 * 64
    qp = np.zeros((BH, tp, hd), np.float32); qp[:, :tokens] = q
    kp = np.zeros((BH, tp, hd), np.float32); kp[:, :tokens] = k
    if kernel.startswith("attn2"):
        assert hd == 64
        variant = int(kernel.split(":")[1]) if ":" in kernel else 0
        vp = np.zeros((BH, tp, hd), np.float32); vp[:, :tokens] = v
        out = np.zeros((1, tokens, BH * hd), np.uint16)
        fn = lib.hiptsdbg_attention2
        fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        _lib.check(fn(_lib.ptr(_bits(qp, f16)), _lib.ptr(_bits(kp, f16)), _lib.ptr(_bits(vp, f16)), _lib.ptr(out), 1, BH, tokens, tp, int(f16), variant, 0, None))
        return _from_bits(out, f16).reshape(tokens, BH, hd).transpose(1, 0, 2)
    vT = np.zeros((BH, hd, tp), np.float32); vT[:, :, :tokens] = v.transpose(0, 2, 1)
    out = np.zeros((1, tokens, BH * hd), np.uint16)
    lib.hiptsdbg_attention_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
    st = lib.hiptsdbg_attention_run(_lib.ptr(_bits(qp, f16)), _lib.ptr(_bits(kp, f16)), _lib.ptr(_bits(vT, f16)), _lib.ptr(out), 1, BH, tokens, tp, hd,
                                    int(f16))
    _lib.check(st)
    return _from_bits(out, f16).reshape(tokens, BH, hd).transpose(1, 0, 2)


def _reference(q, k, v):
    s = np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64))          # log2-domain scores
    s -= s.max(axis=2, keepdims=True)
    p = np.exp2(s)
    return np.einsum("bqk,bkd->bqd", p / p.sum(axis=2, keepdims=True), v.astype(np.float64))


TOL = {0: 2e-2, 1: 3e-3}      # bf16 P and output: 2^-8 relative on O(1) values; IEEE half: 2^-11


# 16: one tile cut to its first half; 64 / 128: no masked tail; 96: the last tile holds exactly 32 keys (half); 97: 33 (masked, both halves);
# 784 / 1025: waves past the last query row (three of 28 / of 36) that only stage
@pytest.mark.parametrize("f16", [0, 1])
@pytest.mark.parametrize("tokens,hd", [(784, 64), (1025, 64), (144, 32), (50, 64), (16, 64), (64, 64), (96, 64), (97, 64), (128, 32), (200, 64)])
def test_attention_matches_float64_softmax(tokens, hd, f16):
    rng = np.random.default_rng(tokens + hd)
    BH = 6
    q = _op(rng.standard_normal((BH, tokens, hd)) * 0.6, f16)       # scores of a few units, like the ViT's
    k = _op(rng.standard_normal((BH, tokens, hd)), f16)
    v = _op(rng.standard_normal((BH, tokens, hd)), f16)
    want = _reference(q, k, v)
    for kernel in (KERNELS64 if hd == 64 else ["attn"]):
        got = _run(q, k, v, tokens, hd, f16, kernel)
        err = np.abs(got - want).max()
        print("%s %dx%d f16=%d: max |error| %.3e" % (kernel, tokens, hd, f16, err))
        assert err <= TOL[f16], kernel


@pytest.mark.parametrize("f16", [0, 1])
def test_attention_one_wave_per_simd_kernel_walks_several_items(f16):
    """csrc/attn3.h (variant 6) is persistent: with more (image, head) chunks than compute units a workgroup takes several items in turn, the K / V
    ring running on across them (the last tile steps of one item stage the first tiles of the next), and a fallback pass in the MIDDLE of a
    workgroup's items (head 3: a late +300 spike) has to leave the ring usable for the items after it."""
    rng = np.random.default_rng(11 + f16)
    BH, tokens, hd = 288, 784, 64                   # 576 items for at most 256 workgroups
    q = _op(rng.standard_normal((BH, tokens, hd)) * 0.6, f16)
    k = _op(rng.standard_normal((BH, tokens, hd)), f16)
    v = _op(rng.standard_normal((BH, tokens, hd)), f16)
    _spike(q, k, 3, 5, 600, 300.0, f16)
    got = _run(q, k, v, tokens, hd, f16, "attn2:6")
    assert np.isfinite(got).all()
    for b in (0, 1, 2, 3, 4, 127, 128, 129, 200, 255, 256, 257, 287):      # float64 reference of a few heads ...
        e = np.abs(got[b] - _reference(q[b:b + 1], k[b:b + 1], v[b:b + 1])[0]).max()
        assert e <= TOL[f16], (b, e)
    base = _run(q, k, v, tokens, hd, f16, "attn2")                          # ... and all of them against the default kernel
    assert np.abs(got - base).max() <= 2 * TOL[f16]


def _spike(q, k, b, row, key, score, f16):
    """Make key `key` of head b score `score` (log2 domain) against query `row`, without touching the other rows' ordinary scores much:
    the key points along sign(q[row]) and the query keeps its sign pattern at a fixed magnitude."""
    hd = q.shape[2]
    sg = np.where(q[b, row] >= 0, 1.0, -1.0).astype(np.float32)
    q[b, row] = _op(sg * 0.5, f16)
    k[b, key] = _op(sg * (score / (0.5 * hd)), f16)
    return float(q[b, row].astype(np.float64) @ k[b, key].astype(np.float64))


@pytest.mark.parametrize("kernel", KERNELS64)
@pytest.mark.parametrize("f16", [0, 1])
def test_attention_fallback_when_scores_leave_the_fast_window(f16, kernel):
    """Rows whose unnormalised sum overflows (a key with a score of +300), underflows (every score below -160) or mixes both ends
    must come out right: the fast path detects them (row sum outside its window: [2^-100, 2^100] with bf16 operands, [2^-11, 2^15]
    relative to the first keys' maximum + 10 with half operands) and the workgroup reruns classically.
    The cases sit in different 128-row query blocks, the rest of the rows take the fast path in the same launch."""
    rng = np.random.default_rng(7)
    BH, tokens, hd = 4, 784, 64
    q = _op(rng.standard_normal((BH, tokens, hd)) * 0.6, f16)
    k = _op(rng.standard_normal((BH, tokens, hd)), f16)
    v = _op(rng.standard_normal((BH, tokens, hd)), f16)
    # head 0, query 5: aligned with key 600 at score ~ +300 (tile 9: the jump comes late, after eight ordinary tiles)
    s05 = _spike(q, k, 0, 5, 600, 300.0, f16)
    # head 1, queries 300..303: every score around -200 (keys all point away)
    base = _op(np.ones(hd) * 1.5, f16)
    k[1] = _op(-base[None, :] * (1.0 + 0.05 * rng.standard_normal((tokens, 1))), f16)
    q[1, 300:304] = base * 1.4
    # head 2, query 700: both a +200 (key 3: inside the first tile) and the ordinary keys
    _spike(q, k, 2, 700, 3, 200.0, f16)
    s1 = float(q[1, 300].astype(np.float64) @ k[1, 0].astype(np.float64))
    assert s05 > 250 and s1 < -160, (s05, s1)
    got = _run(q, k, v, tokens, hd, f16, kernel)
    want = _reference(q, k, v)
    assert np.isfinite(got).all()
    for (b, r) in [(0, 5), (1, 300), (1, 303), (2, 700)]:
        e = np.abs(got[b, r] - want[b, r]).max()
        print("forced-fallback row (%d, %d) f16=%d: max |error| %.3e" % (b, r, f16, e))
        assert e <= TOL[f16]
    assert np.abs(got - want).max() <= TOL[f16]                       # and every other row of the launch
    np.testing.assert_allclose(got[0, 5], v[0, 600], atol=TOL[f16])   # the spike takes all the weight


@pytest.mark.parametrize("kernel", ["attn2", "attn2:3", "attn"])
@pytest.mark.parametrize("f16", [0, 1])
def test_attention_window_edges(f16, kernel):
    """The range in which the fast path stays ACTIVE but P or O could be at risk: scores just inside and just outside each
    window edge, with |V| up to 8 so that O = sum P V is larger than the row sum.
      half: P = 2^(S - m_ref), m_ref = the row's maximum over its first keys (+ 10 bits of head room since round 4: HIPTS_ATTN_REF_MARGIN);
            a later key 13 above that maximum stays on the fast path, 14.9 / 16 sit at the edge of a kernel without head room,
            21 / 22.9 / 24 / 26 around the edge of the one with it (l < 2^15 either way), 60 far outside: none may
            come out as inf (P = 2^16 is +inf in half); a spike INSIDE the first tile at +20 .. +60 is the reference itself; rows
            with every score at or below -30 are ordinary relative to their own first tile.
      bf16: P = 2^S; S = 95 with |v| = 8 is inside (O ~ 2^98 finite), S = 99.5 / 101 / 120 around the 2^100 edge, S = -95 / -101
            / -120 around the lower one."""
    rng = np.random.default_rng(11 + f16)
    BH, tokens, hd = 6, 784, 64
    q = _op(rng.standard_normal((BH, tokens, hd)) * 0.6, f16)
    k = _op(rng.standard_normal((BH, tokens, hd)), f16)
    v = _op(rng.standard_normal((BH, tokens, hd)) * 4.0, f16)
    v[:, 650:660] = _op(np.sign(v[:, 650:660]) * 8.0, f16)
    rows = []
    if f16:
        late = [13.0, 14.9, 16.0, 22.9, 24.9, 26.0, 60.0]
        for i, sc in enumerate(late):                 # late spikes, one per 128-row query block of head 0 (key 650 + i: tile 10)
            r = 128 * i + 7
            _spike(q, k, 0, r, 650 + i, 1.0, f16)                       # fixes the query row; then aim `sc` above ITS first-tile maximum
            m0 = float((q[0, r].astype(np.float64) @ k[0, :64].astype(np.float64).T).max())
            _spike(q, k, 0, r, 650 + i, m0 + sc, f16)
            rows.append((0, r))
        for i, sc in enumerate([20.0, 40.0, 60.0]):   # early spikes (key 3 + i: first tile) -- they ARE the reference
            _spike(q, k, 1, 128 * i + 9, 3 + i, sc, f16)
            rows.append((1, 128 * i + 9))
        base = _op(np.ones(hd), f16)                  # head 2: every score of rows 200..203 between -30 and -40
        k[2] = _op(-base[None, :] * (0.5 + 0.04 * rng.random((tokens, 1))), f16)
        q[2, 200:204] = base
        rows += [(2, 200), (2, 203)]
    else:
        for i, sc in enumerate([95.0, 99.5, 101.0, 120.0]):
            _spike(q, k, 0, 128 * i + 7, 650 + i, sc, f16)
            rows.append((0, 128 * i + 7))
        for i, sc in enumerate([-95.0, -101.0, -120.0]):               # every score of one row near the lower edge
            base = _op(np.ones(hd), f16)
            k[1 + i] = _op(-base[None, :] * ((-sc / hd) * (1.0 + 0.01 * rng.random((tokens, 1)))), f16)
            q[1 + i, 300:302] = base
            rows += [(1 + i, 300)]
    got = _run(q, k, v, tokens, hd, f16, kernel)
    want = _reference(q, k, v)
    assert np.isfinite(got).all()
    scale = np.abs(want).max(axis=2, keepdims=True) + 1.0
    for (b, r) in rows:
        e = (np.abs(got[b, r] - want[b, r]) / scale[b, r]).max()
        print("window-edge row (%d, %d) f16=%d: max relative error %.3e" % (b, r, f16, e))
        assert e <= TOL[f16]
    assert (np.abs(got - want) / scale).max() <= TOL[f16]


@pytest.mark.parametrize("kernel", ["attn2", "attn"])
def test_attention_classic_path_agrees(kernel):
    """HIPTS_ATTN_CLASSIC=1 (per-tile running maximum everywhere) and the default fast path give the same output up to bf16 rounding."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_attention import _run, _bf16
rng = np.random.default_rng(3)
q = _bf16(rng.standard_normal((3, 784, 64)) * 0.6); k = _bf16(rng.standard_normal((3, 784, 64))); v = _bf16(rng.standard_normal((3, 784, 64)))
np.save(sys.argv[1], _run(q, k, v, 784, 64, 0, sys.argv[2]))
""" % (ROOT, os.path.join(ROOT, "anime-illust-image-searcher_amd"), os.path.join(ROOT, "tests"))
    import tempfile
    outs = []
    for classic in ("0", "1"):
        f = tempfile.mktemp(suffix=".npy")
        r = subprocess.run([sys.executable, "-c", code, f, kernel], capture_output=True, text=True, env=dict(os.environ, HIPTS_ATTN_CLASSIC=classic), timeout=300)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        outs.append(np.load(f))
        os.remove(f)
    assert np.abs(outs[0] - outs[1]).max() <= 1.6e-2                  # two bf16 roundings apart at most
