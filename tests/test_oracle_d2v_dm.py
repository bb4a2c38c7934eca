"""CPU: the PV-DM inference oracle (oracle/csrc/oracle.c::orc_d2v_infer_dm) against a step followed by hand in numpy.

PARITY UNPINNED like the PV-DBOW oracle (gensim absent): this pins the C restatement to the published algorithm
(doc2vec_inner.pyx::train_document_dm / fast_document_dm_neg) evaluated independently for a document small enough to
follow -- two words, no sub-sampling, no negative samples, one epoch, a window that covers the document whatever the
reduced windows are -- in both dm_mean modes."""
import numpy as np

from oracle import d2v as od2v


def _dot_wave64(v, w):
    """the dot-product order of the oracle and the HIP kernel: 64 lane partials (fma chains over l, l + 64, ...), xor butterfly"""
    p = np.zeros(64, dtype=np.float32)
    for l in range(64):
        acc = np.float32(0.0)
        for i in range(l, len(v), 64):
            acc = np.float32(np.float64(v[i]) * np.float64(w[i]) + np.float64(acc))      # fmaf: one rounding (the product of two floats is exact in double)
        p[l] = acc
    m = 32
    while m >= 1:
        p = (p + p[np.arange(64) ^ m]).astype(np.float32)
        m >>= 1
    return p[0]


def _fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def test_dm_first_epoch_by_hand():
    rng = np.random.default_rng(3)
    V, dim = 6, 100
    syn1neg = (rng.standard_normal((V, dim)) * 0.2).astype(np.float32)
    wv = (rng.standard_normal((V, dim)) * 0.3).astype(np.float32)
    cum = np.array([10, 20, 30, 40, 50, 2 ** 31 - 1], dtype=np.uint32)
    v0 = (rng.standard_normal((1, dim)) * 0.01).astype(np.float32)
    ptr = np.array([0, 2], dtype=np.int64)
    words = np.array([4, 1], dtype=np.int32)
    seeds = np.array([12345], dtype=np.uint64)
    table = od2v.exp_table()
    alpha = np.float32(0.025)
    for dm_mean in (1, 0):
        got = od2v.infer_dm(syn1neg, wv, cum, None, ptr, words, v0, seeds, epochs=1, alpha=0.025, min_alpha=1e-4, negative=0, exp_scale=83.0,
                            window=5, dm_mean=dm_mean)[0]
        v = v0[0].copy()
        for i, (w, other) in enumerate(((4, 1), (1, 4))):
            l1 = (np.zeros(dim, np.float32) + wv[other]).astype(np.float32)
            l1 = (l1 + v).astype(np.float32)
            inv = np.float32(1.0) / np.float32(2.0)
            if dm_mean:
                l1 = (l1 * inv).astype(np.float32)
            f = _dot_wave64(l1, syn1neg[w])
            assert -6 < f < 6
            fs = table[int(np.float64(np.float32(f + np.float32(6.0))) * 83.0)]
            g = np.float32(np.float32(np.float32(1.0) - fs) * alpha)
            work = np.array([_fma32(g, syn1neg[w][c], np.float32(0.0)) for c in range(dim)], dtype=np.float32)
            if not dm_mean:
                work = (work * inv).astype(np.float32)
            v = (v + work).astype(np.float32)
        assert got.tobytes() == v.tobytes(), dm_mean


def test_dm_properties():
    """deterministic; the window matters once it is shorter than the document; out-of-vocabulary words are dropped"""
    rng = np.random.default_rng(4)
    V, dim, n = 50, 64, 12
    syn1neg = (rng.standard_normal((V, dim)) * 0.2).astype(np.float32)
    wv = (rng.standard_normal((V, dim)) * 0.3).astype(np.float32)
    cum = np.cumsum(rng.integers(1, 100, V)).astype(np.float64)
    cum = np.round(cum / cum[-1] * (2 ** 31 - 1)).astype(np.uint32)
    v0 = (rng.standard_normal((1, dim)) * 0.01).astype(np.float32)
    words = rng.integers(0, V, n).astype(np.int32)
    ptr = np.array([0, n], dtype=np.int64)
    seeds = np.array([7], dtype=np.uint64)
    a = od2v.infer_dm(syn1neg, wv, cum, None, ptr, words, v0, seeds, 5, window=50)
    b = od2v.infer_dm(syn1neg, wv, cum, None, ptr, words, v0, seeds, 5, window=50)
    c = od2v.infer_dm(syn1neg, wv, cum, None, ptr, words, v0, seeds, 5, window=2)
    assert a.tobytes() == b.tobytes() and not np.array_equal(a, c) and np.isfinite(a).all()
    w2 = np.concatenate([words[:5], [-1, V + 3], words[5:]]).astype(np.int32)
    d = od2v.infer_dm(syn1neg, wv, cum, None, np.array([0, n + 2], dtype=np.int64), w2, v0, seeds, 5, window=50)
    assert d.tobytes() == a.tobytes()
