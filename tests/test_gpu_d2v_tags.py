"""GPU parity: Doc2Vec PV-DBOW inference (bit-exact vs the C oracle with the same explicit inputs)
and tag selection (index-exact vs the numpy oracle and the reference-captured golden lines)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,V,ndocs,epochs,with_sample", [(300, 2000, 300, 20, True), (300, 2000, 64, 100, False),
                                                            (64, 500, 100, 7, True), (100, 70, 50, 5, True),
                                                            (64, 500, 8, 3, True)])      # the shape a round-2 bring-up build of the planned kernel faulted on (tools/gpurun/r2_dbg.sh)
def test_d2v_infer_bit_exact(dim, V, ndocs, epochs, with_sample):
    from hiptagsearch import synth
    from hiptagsearch.d2v import Doc2VecInference
    from oracle import d2v as od2v
    ptr, terms = synth.tag_corpus(D=ndocs, V=V, seed=7)
    terms = terms.copy()
    terms[::17] = -1                                         # out-of-vocabulary tokens are dropped
    counts = synth.term_counts(ptr, terms, V)
    m = synth.d2v_model(counts, dim=dim, seed=44)
    v0, seeds = synth.d2v_inputs(ndocs, dim, seed=44)
    si = m["sample_int"] if with_sample else None
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], si, {}, epochs=epochs)
    got = model.infer_batch(ptr, terms, v0, seeds)
    want = od2v.infer(m["syn1neg"], m["cum_table"], si, ptr, terms, v0, seeds, epochs)
    assert np.isfinite(got).all()
    assert got.tobytes() == want.tobytes()
    assert not np.array_equal(got, v0)


@pytest.mark.parametrize("dim,V,ndocs,epochs,window,dm_mean,with_sample", [(300, 2000, 120, 20, 50, 1, True), (300, 2000, 40, 100, 5, 0, False),
                                                                             (64, 500, 100, 7, 3, 1, True), (100, 70, 50, 5, 50, 0, True)])
def test_d2v_infer_dm_bit_exact(dim, V, ndocs, epochs, window, dm_mean, with_sample):
    """PV-DM inference (dm=1: mean or sum of context word vectors + document vector; the form BASELINE.json's north_star names) against
    the C oracle with the same explicit inputs -- document vectors bit for bit.  window 50 is the reference model's (genmodel.py:159),
    3 and 5 cut the context inside a document so the reduced windows matter."""
    from hiptagsearch import synth
    from hiptagsearch.d2v import Doc2VecInference
    from oracle import d2v as od2v
    ptr, terms = synth.tag_corpus(D=ndocs, V=V, seed=9)
    terms = terms.copy()
    terms[::13] = -1
    counts = synth.term_counts(ptr, terms, V)
    m = synth.d2v_model(counts, dim=dim, seed=45)
    wv = (np.random.default_rng(46).standard_normal((V, dim)) * 0.3).astype(np.float32)      # a trained model's word vectors: O(0.1 .. 1)
    v0, seeds = synth.d2v_inputs(ndocs, dim, seed=45)
    si = m["sample_int"] if with_sample else None
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], si, {}, epochs=epochs, dm=1, word_vectors=wv, window=window, dm_mean=dm_mean)
    got = model.infer_batch(ptr, terms, v0, seeds)
    want = od2v.infer_dm(m["syn1neg"], wv, m["cum_table"], si, ptr, terms, v0, seeds, epochs, window=window, dm_mean=dm_mean)
    assert np.isfinite(got).all()
    assert got.tobytes() == want.tobytes()
    assert not np.array_equal(got, v0)
    # and it is a different model from PV-DBOW on the same arrays
    dbow = Doc2VecInference(m["syn1neg"], m["cum_table"], si, {}, epochs=epochs).infer_batch(ptr, terms, v0, seeds)
    assert not np.array_equal(got, dbow)


def test_d2v_infer_dm_refuses_what_it_cannot_run():
    import hiptagsearch
    from hiptagsearch import synth
    from hiptagsearch.d2v import Doc2Vec, Doc2VecInference
    V, dim = 100, 64
    ptr, terms = synth.tag_corpus(D=4, V=V, seed=1)
    m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=dim)
    with pytest.raises(ValueError):
        Doc2VecInference(m["syn1neg"], m["cum_table"], None, {}, dm=1)                      # no word vectors
    with pytest.raises(NotImplementedError):
        Doc2Vec(vector_size=dim, dm=1)                                                      # training stays PV-DBOW
    wv = np.zeros((V, dim), np.float32)
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], None, {}, epochs=2, dm=1, word_vectors=wv, window=5, dm_mean=1)
    long_ptr = np.array([0, 600], dtype=np.int64)
    with pytest.raises(hiptagsearch.HipTagSearchError):
        model.infer_batch(long_ptr, np.zeros(600, np.int32), np.zeros((1, dim), np.float32), np.zeros(1, np.uint64))


def test_d2v_gensim_shaped_interface():
    from hiptagsearch import synth
    from hiptagsearch.d2v import Doc2VecInference
    V = 300
    ptr, terms = synth.tag_corpus(D=200, V=V, seed=3)
    m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=300)
    toks = synth.vocab_tokens(V)
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], m["sample_int"], {t: i for i, t in enumerate(toks)}, epochs=10)
    a = model.infer_vector([toks[1], toks[5], "unknown-tag"])
    b = model.infer_vector([toks[1], toks[5], "unknown-tag"])
    assert a.shape == (300,) and a.dtype == np.float32
    np.testing.assert_array_equal(a, b)                      # deterministic, unlike gensim's hash()-seeded start
    c = model.infer_vector([toks[2]])
    assert not np.array_equal(a, c)


# --------------------------------------------------------------------------------- tag selection
def test_tagsel_golden_lines(golden_dir):
    """Lines captured from tagging.py's Predictor.predict (fake model, real post-processing)."""
    import torch
    from hiptagsearch.tagger import TagSelector, format_lines
    z = np.load(os.path.join(golden_dir, "g4_predict.npz"))
    g = json.load(open(os.path.join(golden_dir, "g4_predict.json")))
    probs = torch.sigmoid(torch.from_numpy(z["logits"])).numpy()
    sel = TagSelector(z["category"], max_batch=8)
    counts, ids, thr = sel.run(probs)
    assert format_lines(g["names"], counts, ids) == g["lines"]


@pytest.mark.parametrize("C,B", [(10861, 16), (600, 5), (70, 3)])
def test_tagsel_matches_oracle(C, B):
    from hiptagsearch import synth
    from hiptagsearch.tagger import TagSelector
    from oracle import tags as otags
    names, cat = synth.label_table(C)
    rng = np.random.default_rng(C)
    logits = (rng.standard_normal((B, C)) * 3).astype(np.float32)
    logits[0, :] = 0.0                                        # all equal
    logits[1, 10:40] = 9.0                                    # saturated ties
    probs = otags.sigmoid_f32(logits)
    sel = TagSelector(cat, max_batch=B)
    for (gt, gm, ct, cm) in [(0.3, True, 0.3, True), (0.35, False, 0.85, False), (0.3, True, 0.5, False)]:
        counts, ids, thr = sel.run(probs, gt, gm, ct, cm)
        gi, ci = list(np.where(cat == 0)[0]), list(np.where(cat == 4)[0])
        for r in range(B):
            g, c, tg, tc = otags.select_indices(probs[r], gi, ci, gt, gm, ct, cm)
            assert counts[r].tolist() == [len(g), len(c)]
            assert ids[r, :len(g)].tolist() == g
            assert ids[r, len(g):len(g) + len(c)].tolist() == c
            assert float(thr[r, 0]).hex() == float(tg).hex() and float(thr[r, 1]).hex() == float(tc).hex()


def test_tagsel_row_cap_truncates():
    from hiptagsearch import synth
    from hiptagsearch.tagger import TagSelector
    names, cat = synth.label_table(600)
    probs = np.random.default_rng(1).random((2, 600), dtype=np.float32)
    sel = TagSelector(cat, max_batch=2)
    full_counts, full_ids, _ = sel.run(probs)
    counts, ids, _ = sel.run(probs, row_cap=16)
    np.testing.assert_array_equal(counts, full_counts)       # full counts reported, ids truncated
    for r in range(2):
        n = min(16, int(full_counts[r].sum()))
        np.testing.assert_array_equal(ids[r, :n], full_ids[r, :n])
