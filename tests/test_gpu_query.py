"""GPU parity: query-scoring path through the C ABI vs the CPU oracle.
Bit-exact for float64 BM25, float32 index products (k-ordered fma chain) and ranking."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fh(x):
    return float.fromhex(x)


@pytest.fixture(scope="module")
def corpus20k():
    from hiptagsearch import synth
    ptr, terms = synth.tag_corpus(D=20_000, V=3_000, seed=42)
    return ptr, terms, 3_000


def _oracle_index(ptr, terms, V):
    """Build the oracle's BM25 objects from a CSR of token ids (ids are their own dictionary ids)."""
    from oracle import bm25 as obm25
    docs = [[str(t) for t in terms[ptr[d]:ptr[d + 1]]] for d in range(len(ptr) - 1)]
    token2id = {str(i): i for i in range(V)}
    return obm25.bm25_build(docs, token2id)


# --------------------------------------------------------------------------------- BM25
@pytest.mark.parametrize("case", ["tiny", "d1000"])
def test_bm25_golden_bit_exact(golden_dir, case):
    """Reference-captured vectors (genmodel.py / webui.py numpy code) through the HIP kernel."""
    from hiptagsearch.bm25 import BM25Index
    g1 = json.load(open(os.path.join(golden_dir, "g1_bm25_build.json")))[case]
    g2 = json.load(open(os.path.join(golden_dir, "g2_bm25_score.json")))[case]
    idx = BM25Index.from_tokens(g1["docs"], g1["token2id"])
    corpus, idf, avgdl, D, dl = idx.reference_objects()
    assert [{str(k): v for k, v in d.items()} for d in corpus] == g1["corpus"]
    assert list(idf.keys()) == g1["idf_keys"]
    assert [float(v).hex() for v in idf.values()] == g1["idf_hex"]
    assert float(avgdl).hex() == g1["avgdl_hex"] and D == g1["D"]
    assert dl.tolist() == g1["doc_lengths"] and str(dl.dtype) == g1["doc_lengths_dtype"]
    queries = [{int(k): float(v) for k, v in q} for q in g2["queries"]]
    got = idx.score(queries)
    for i, want_hex in enumerate(g2["scores_hex"]):
        want = np.array([fh(x) for x in want_hex])
        assert got[i].tobytes() == want.tobytes(), "query %d" % i


def test_bm25_20k_bit_exact(corpus20k):
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from oracle import bm25 as obm25
    ptr, terms, V = corpus20k
    idx = BM25Index(ptr, terms, V)
    corpus, idf, avgdl, D, dl = _oracle_index(ptr, terms, V)
    e = idx.export()
    optr, oterm, otf = obm25.to_csr(corpus)
    np.testing.assert_array_equal(e["csr_ptr"], optr)
    np.testing.assert_array_equal(e["csr_term"], oterm)
    np.testing.assert_array_equal(e["csr_tf"], otf)
    np.testing.assert_array_equal(e["doc_len"], dl)
    assert float(idx.avgdl).hex() == float(avgdl).hex()
    idf_arr = np.zeros(V)
    for k, v in idf.items():
        idf_arr[k] = v
    assert e["idf"].tobytes() == idf_arr.tobytes()          # numpy-evaluated idf installed by the wrapper
    qs = synth.queries(64, V, seed=43, head=500)
    got = idx.score([dict(q) for q in qs])
    for i, q in enumerate(qs):
        want = obm25.bm25_score_csr(optr, oterm, otf, idf_arr, avgdl, dl, [t for t, _ in q], [w for _, w in q])
        assert got[i].tobytes() == want.tobytes(), "query %d: %r" % (i, q)


def test_bm25_libm_idf_within_one_ulp(corpus20k):
    from hiptagsearch.bm25 import BM25Index
    ptr, terms, V = corpus20k
    a = BM25Index(ptr, terms, V, numpy_idf=False).export()["idf"]
    b = BM25Index(ptr, terms, V, numpy_idf=True).export()["idf"]
    nz = b != 0
    assert np.max(np.abs(a[nz] - b[nz]) / b[nz]) <= 2.3e-16      # tolerance: 1 ulp of float64


# --------------------------------------------------------------------------------- similarity
# more than 32 queries take sim_mfma_wide_kernel (up to 256 queries per pass over the index, 2 / 4 / 8 query blocks per wave): ragged
# document and query counts, K % 8 == 4 and == 0, one and two wide passes, a 32-query remainder on the narrow kernel
@pytest.mark.parametrize("D,K,nq", [(1000, 300, 1), (4099, 300, 5), (3000, 768, 33), (64, 8, 2), (31, 12, 1),
                                    (5, 300, 40), (40, 4, 64), (1000, 300, 100), (4099, 300, 256), (777, 12, 300), (100, 8, 289), (100, 8, 288)])
def test_similarity_bit_exact(D, K, nq):
    from hiptagsearch.index import Similarity
    from oracle import search as osearch
    rng = np.random.default_rng(D + K)
    rows = rng.standard_normal((D, K)).astype(np.float32)
    q = rng.standard_normal((nq, K)).astype(np.float32)
    idx = Similarity("t", None, K)
    idx.add_matrix(rows[: D // 2])
    idx.add_matrix(rows[D // 2:])                       # exercises growth
    assert len(idx) == D
    np.testing.assert_array_equal(idx.vector_by_id(D - 1), rows[D - 1])
    got = idx.query(q)
    for i in range(nq):
        want = osearch.similarity(rows, q[i])
        assert got[i].tobytes() == want.tobytes(), "query %d" % i


def test_similarity_gensim_forms():
    """Sparse list-of-tuples documents are unit-normalised on add (gen_cfeatures.py:310-314), ndarray
    documents are stored as given (genmodel.py:171-173)."""
    from hiptagsearch.index import Similarity
    v = np.arange(1, 9, dtype=np.float32)
    a = Similarity("a", [v], 8)
    b = Similarity("b", [[(i, float(x)) for i, x in enumerate(v)]], 8)
    np.testing.assert_array_equal(a.vector_by_id(0), v)
    np.testing.assert_allclose(np.linalg.norm(b.vector_by_id(0)), 1.0, rtol=1e-6)
    a.add_documents([v * 2])
    assert len(a) == 2


# --------------------------------------------------------------------------------- top-k
def _rank_oracle(vals, k):
    from oracle import search as osearch
    return osearch.topk(vals, k)


@pytest.mark.parametrize("n,k", [(100_000, 100), (100_000, 1024), (5000, 800), (70, 100), (1, 1)])
def test_topk_random(n, k):
    import torch
    from hiptagsearch import _lib
    rng = np.random.default_rng(n + k)
    vals = rng.random((3, n))
    vals[1, rng.integers(0, n, n // 3)] = -np.inf
    vals[2] = np.round(vals[2], 2)                       # massive exact ties
    dev = torch.from_numpy(vals).cuda()
    ids = np.empty((3, k), np.int32)
    out = np.empty((3, k), np.float64)
    _lib.call("hipts_topk", _lib.ptr(dev), 3, _lib.c_int64(n), k, _lib.ptr(ids), _lib.ptr(out), _lib.HOST, 0, None)
    kk = min(k, n)
    for r in range(3):
        wi, wv = _rank_oracle(vals[r], kk)
        np.testing.assert_array_equal(ids[r, :kk], wi)
        assert out[r, :kk].tobytes() == wv.tobytes()
        assert (ids[r, kk:] == -1).all()


def test_topk_all_equal_and_all_inf():
    import torch
    from hiptagsearch import _lib
    n, k = 50_000, 700
    vals = np.stack([np.full(n, 0.25), np.full(n, -np.inf), np.concatenate([np.zeros(n - 5), -np.zeros(5)])])
    dev = torch.from_numpy(vals).cuda()
    ids = np.empty((3, k), np.int32)
    out = np.empty((3, k), np.float64)
    _lib.call("hipts_topk", _lib.ptr(dev), 3, _lib.c_int64(n), k, _lib.ptr(ids), _lib.ptr(out), _lib.HOST, 0, None)
    for r in range(3):
        np.testing.assert_array_equal(ids[r], np.arange(k))      # all tied -> ascending doc id


# --------------------------------------------------------------------------------- fused search
def test_search_rank_equal(corpus20k):
    """hipts_search (BM25 + index product + normalise + combine + top-k) vs the oracle restatement of
    webui.py:352-383,191-192: identical ids, bit-identical scores."""
    import torch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25
    from oracle import search as osearch
    ptr, terms, V = corpus20k
    D = len(ptr) - 1
    rows = synth.index_vectors(D, 300, seed=46)
    bm = BM25Index(ptr, terms, V)
    index = Similarity("idx", None, 300, capacity=D)
    index.add_matrix(rows)
    eng = SearchEngine(None, index, {}, bm, [])
    corpus, idf, avgdl, _, dl = _oracle_index(ptr, terms, V)
    optr, oterm, otf = obm25.to_csr(corpus)
    idf_arr = np.zeros(V)
    for kk_, v in idf.items():
        idf_arr[kk_] = v
    qs = synth.queries(40, V, seed=44, head=300)
    rng = np.random.default_rng(5)
    qv = rng.standard_normal((len(qs), 300))
    qv /= np.linalg.norm(qv, axis=1, keepdims=True)
    k = 100
    final_dev = torch.empty((len(qs), D), dtype=torch.float64, device="cuda")
    ids, vals = eng.score_topk([dict(q) for q in qs], qv, k, final_out=final_dev)
    final_host = final_dev.cpu().numpy()
    for i, q in enumerate(qs):
        b = obm25.bm25_score_csr(optr, oterm, otf, idf_arr, avgdl, dl, [t for t, _ in q], [w for _, w in q])
        s = osearch.similarity(rows, qv[i].astype(np.float32))
        final = osearch.combine(b, s)
        assert final_host[i].tobytes() == final.tobytes(), "query %d combined scores" % i
        wi, wv = osearch.topk(final, k)
        np.testing.assert_array_equal(ids[i], wi)
        assert vals[i].tobytes() == wv.tobytes()


@pytest.mark.parametrize("D,V", [(300, 60), (9000, 800), (20000, 2000)])
def test_search_without_stored_rows_equals_search_with_them(D, V):
    """hipts_search for a batch, final_out absent: the top-k kernel combines BM25 and index scores where it reads them (no combine launch,
    no stored rows).  Same ids and bit-identical scores as the call that stores the combined rows -- the form test_search_rank_equal pins to
    the oracle -- for k below / above the candidate capacity, plain / required / excluded terms, indexes below and above the fast path's
    8192-document floor (the exact radix select reads its scores through the same expression)."""
    import torch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    ptr, terms = synth.tag_corpus(D, V, seed=7)
    bm = BM25Index(ptr, terms, V)
    index = Similarity("idx", None, 300, capacity=D)
    index.add_matrix(synth.index_vectors(D, 300, seed=8))
    eng = SearchEngine(None, index, {}, bm, [])
    qs = [dict(q) for q in synth.queries(70, V, seed=9, head=min(V, 300))]
    qv = np.random.default_rng(3).standard_normal((len(qs), 300)).astype(np.float32)
    for k in (5, 100, min(1024, D)):
        final_dev = torch.empty((len(qs), D), dtype=torch.float64, device="cuda")
        ids_a, vals_a = eng.score_topk(qs, qv, k, final_out=final_dev)
        ids_b, vals_b = eng.score_topk(qs, qv, k)
        np.testing.assert_array_equal(ids_b, ids_a, err_msg="D %d k %d" % (D, k))
        assert vals_b.tobytes() == vals_a.tobytes(), (D, k)


def test_search_one_query_path_is_bit_equal_to_the_batched_path(corpus20k):
    """hipts_search with nq == 1 takes the one-query path (thread-per-document BM25 in the reference's document-major form,
    fmaf-chain index product, sampled threshold + candidate ranking): ids, scores and the combined score row must equal the
    batched kernels' bit for bit -- plain, required and excluded terms, k = 100 and k = 1024, and the cases that defeat the
    sampled threshold (everything -inf, massive exact ties), which fall through to the exact radix select."""
    import ctypes
    import torch
    from hiptagsearch import _lib, synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25
    from oracle import search as osearch
    ptr, terms, V = corpus20k
    D = len(ptr) - 1
    rows = synth.index_vectors(D, 300, seed=46)
    rows[5000:9000] = rows[4999]                                   # 4001 identical rows: exact ties in the index product
    bm = BM25Index(ptr, terms, V)
    index = Similarity("idx", None, 300, capacity=D)
    index.add_matrix(rows)
    eng = SearchEngine(None, index, {}, bm, [])
    qs = [dict(q) for q in synth.queries(40, V, seed=44, head=300)]
    qs.append({7: 1.0, V + 5: 1001.0})                             # a required term nobody has: every score -inf (webui.py:161-168)
    qs.append({})                                                  # no BM25 term at all: the index product alone, ties included
    rng = np.random.default_rng(5)
    qv = rng.standard_normal((len(qs), 300))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    for k in (100, 1024):
        fb = torch.empty((len(qs), D), dtype=torch.float64, device="cuda")
        bi, bv = eng.score_topk(qs, qv, k, final_out=fb)           # batched kernels
        fb = fb.cpu().numpy()
        _lib.call("hipts_query_profile_enable", bm._h, 1)
        for i, q in enumerate(qs):
            f1 = torch.empty((1, D), dtype=torch.float64, device="cuda")
            oi, ov = eng.score_topk([q], qv[i:i + 1], k, final_out=f1)
            assert f1.cpu().numpy()[0].tobytes() == fb[i].tobytes(), "query %d combined scores" % i
            np.testing.assert_array_equal(oi[0], bi[i], err_msg="query %d k %d" % (i, k))
            assert ov[0].tobytes() == bv[i].tobytes()
        ms, n, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.call("hipts_query_profile_read", bm._h, 5, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
        assert n.value == len(qs) and ms.value > 0 and by.value > D * 300 * 4 * len(qs)      # the one-query path really ran
        _lib.call("hipts_query_profile_enable", bm._h, 0)
    # and against the oracle directly (not only against the other device path)
    corpus, idf, avgdl, _, dl = _oracle_index(ptr, terms, V)
    optr, oterm, otf = obm25.to_csr(corpus)
    idf_arr = np.zeros(V)
    for t, v in idf.items():
        idf_arr[t] = v
    for i in (0, 3, 17, 40, 41):
        q = qs[i]
        b = obm25.bm25_score_csr(optr, oterm, otf, idf_arr, avgdl, dl, list(q.keys()), list(q.values()))
        f = osearch.combine(b, osearch.similarity(rows, qv[i]))
        wi, wv = osearch.topk(f, 100)
        gi, gv = eng.score_topk([q], qv[i:i + 1], 100)
        np.testing.assert_array_equal(gi[0], wi)
        assert gv[0].tobytes() == wv.tobytes()


def test_search_one_query_many_in_a_row(corpus20k):
    """The default one-query path (results stored to pinned memory by the last kernel and published by a sequence number the host
    spins on; per-query state -- maxima slots, group maxima, candidate slots -- reused from call to call): 300 different queries back
    to back, plain and masked, k = 100 and 1024, against the batched kernels.  State leaking from one query into the next would show
    here.  (At 20 k documents many of these take the exact select inside the ranking kernel: both ways are covered.)"""
    import ctypes
    from hiptagsearch import _lib, synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    ptr, terms, V = corpus20k
    D = len(ptr) - 1
    rows = synth.index_vectors(D, 300, seed=47)
    bm = BM25Index(ptr, terms, V)
    index = Similarity("idx", None, 300, capacity=D)
    index.add_matrix(rows)
    eng = SearchEngine(None, index, {}, bm, [])
    qs = [dict(q) for q in synth.queries(300, V, seed=45, head=300)]
    for i in range(0, 300, 7):                                     # every seventh query requires one of its own terms
        if qs[i]:
            t = next(iter(qs[i]))
            qs[i][t] = 1001.0
    rng = np.random.default_rng(6)
    qv = rng.standard_normal((len(qs), 300))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    for k in (100, 1024):
        bi, bv = eng.score_topk(qs, qv, k)                         # batched kernels
        decided = 0
        for i, q in enumerate(qs):
            oi, ov = eng.score_topk([q], qv[i:i + 1], k)
            np.testing.assert_array_equal(oi[0], bi[i], err_msg="query %d k %d" % (i, k))
            assert ov[0].tobytes() == bv[i].tobytes(), "query %d k %d scores" % (i, k)
            c, fast = ctypes.c_uint32(), ctypes.c_uint32()
            _lib.call("hiptsdbg_search1_last", bm._h, ctypes.byref(c), ctypes.byref(fast))
            decided += fast.value
        assert 0 <= decided <= len(qs)


# --------------------------------------------------------------------------------- full query function
def test_find_similar_documents_matches_oracle():
    """find_similar_documents (webui.py:345-390, normal mode incl. the 10-document rerank and the gap
    filter) against the oracle restatement; Doc2Vec vectors come from the C oracle with the same
    explicit start vectors / seeds the product derives."""
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.d2v import Doc2VecInference, pseudorandom_weak_vector
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25, d2v as od2v, search as osearch
    V, D, dim, epochs = 400, 3000, 300, 8
    ptr, terms = synth.tag_corpus(D=D, V=V, seed=11)
    toks = synth.vocab_tokens(V)
    docs = [[toks[t] for t in terms[ptr[d]:ptr[d + 1]]] for d in range(D)]
    lines = ["img%05d.png," % d + ",".join(docs[d]) for d in range(D)]
    token2id = {t: i for i, t in enumerate(toks)}
    m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=dim, seed=44)
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], m["sample_int"], token2id, epochs=epochs)

    def oracle_infer(list_of_docs):
        p = np.zeros(len(list_of_docs) + 1, dtype=np.int64)
        ids = []
        for i, d in enumerate(list_of_docs):
            ids.extend(token2id.get(t, -1) for t in d)
            p[i + 1] = len(ids)
        v0 = np.stack([pseudorandom_weak_vector(dim, " ".join(d)) for d in list_of_docs])
        seeds = np.asarray([model._seed_for(d) for d in list_of_docs], dtype=np.uint64)
        return od2v.infer(m["syn1neg"], m["cum_table"], m["sample_int"], p, np.asarray(ids, np.int32), v0, seeds, epochs)

    rows = oracle_infer(docs)
    got_rows = model.infer_vectors(docs)
    assert got_rows.tobytes() == rows.tobytes()                    # genmodel.py:168-169 on the device == oracle
    index = Similarity("idx", None, dim, capacity=D)
    index.add_matrix(rows)
    bm = BM25Index.from_tokens(docs, token2id)
    eng = SearchEngine(model, index, token2id, bm, lines)
    corpus, idf, avgdl, _, dl = obm25.bm25_build(docs, token2id)
    for query in [toks[3], "%s %s:+2" % (toks[1], toks[7]), "%s:-1 %s %s:3" % (toks[0], toks[5], toks[9]), toks[2] + ":+1"]:
        got = eng.find_similar_documents(query, topn=50)
        d2v_terms, allw, bm_terms = osearch.parse_query(query)
        qvec = osearch.query_vector(d2v_terms, allw, lambda words: oracle_infer([words])[0], dim)
        sims = osearch.similarity(rows, qvec.astype(np.float32))
        b = obm25.bm25_score(corpus, idf, avgdl, D, dl, osearch.query_weights(bm_terms, token2id))
        final = osearch.combine(b, sims)

        def rerank_sims(top_ids, top_scores):
            vecs = oracle_infer([docs[int(i)] for i in top_ids]).astype(np.float64)
            mean = np.average(vecs, axis=0, weights=top_scores)
            mean = mean / np.linalg.norm(mean)
            return osearch.similarity(rows, mean.astype(np.float32))

        want = osearch.rerank(final, 50, rerank_sims)
        assert [d for d, _ in got] == [d for d, _ in want], query
        np.testing.assert_array_equal(np.array([s for _, s in got]), np.array([s for _, s in want]))
    assert eng.stats["full_rank_fallbacks"] == 0
    print("rank continuations:", eng.stats["rank_continuations"])


def test_rerank_continues_past_rank_1024_on_the_device():
    """The gap filter's second cut point far down the list (ranks ~1500 and ~2000 of 2500, no near-tie before them): round 2 ranked all
    scores on the host for such a query; now the device ranking is continued 1024 entries at a time.  Same result as the oracle's full
    sort, no host fallback, and the continuation really ran."""
    import torch
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import search as osearch
    D, dim = 2500, 4
    a = (1.0 - 3e-4 * np.arange(D)).astype(np.float32)            # spacing 3e-4 >> 1e-6 after the 0.3 weight and the normalisation
    a[1500] = np.nextafter(a[1499], np.float32(0))                # two planted near-ties (one float32 step apart)
    a[2000] = np.nextafter(a[1999], np.float32(0))
    perm = np.random.default_rng(3).permutation(D)
    rows = np.zeros((D, dim), np.float32)
    rows[perm, 0] = a                                             # document perm[i] has rerank similarity a[i]

    class StubModel:                                              # every document infers to e0: the rerank query is e0, rs = rows[:, 0]
        vector_size = dim
        def infer_vectors(self, docs):
            v = np.zeros((len(docs), dim), np.float32); v[:, 0] = 1.0
            return v

    index = Similarity("idx", None, dim, capacity=D)
    index.add_matrix(rows)                                        # stored as given (genmodel.py:171-173)
    lines = ["img%05d.png,t" % d for d in range(D)]
    eng = SearchEngine(StubModel(), index, {"t": 0}, None, lines)
    final = np.full(D, 1e-3, np.float64)                          # first-stage scores all equal: its top 10 are documents 0..9
    final_dev = torch.from_numpy(final).cuda().reshape(1, D)
    ids = np.arange(1024, dtype=np.int32); vals = np.full(1024, 1e-3)
    got = eng._doc2vec_rerank(final_dev, ids, vals, topn=5000)
    want = osearch.rerank(final, 5000, lambda top_ids, top_scores: osearch.similarity(rows, np.array([1, 0, 0, 0], np.float32)))
    assert [d for d, _ in got] == [d for d, _ in want]
    np.testing.assert_array_equal(np.array([s for _, s in got]), np.array([s for _, s in want]))
    assert eng.stats["full_rank_fallbacks"] == 0 and eng.stats["rank_continuations"] >= 1, eng.stats
    assert len(got) > 1024


def test_topk_after_continues_the_ranking():
    """hipts_topk_after: the ranking 1024 entries at a time equals one full sort by (value descending, index ascending), through
    exact ties and into the -inf tail (what _doc2vec_rerank relies on when the gap filter's second cut point lies past rank 1024)."""
    import ctypes
    import torch
    from hiptagsearch import _lib
    rng = np.random.default_rng(12)
    n = 5000
    v = np.round(rng.standard_normal(n), 2)                       # many exact ties
    v[rng.choice(n, 1500, replace=False)] = -np.inf               # a masked tail
    dev = torch.from_numpy(v).cuda()
    order = np.lexsort((np.arange(n), -v))
    ids = np.empty((1, 1024), np.int32); vals = np.empty((1, 1024), np.float64)
    _lib.call("hipts_topk", _lib.ptr(dev), 1, ctypes.c_int64(n), 1024, _lib.ptr(ids), _lib.ptr(vals), _lib.HOST, 0, _lib.current_stream_ptr())
    got_ids, got_vals = list(ids[0]), list(vals[0])
    while len(got_ids) < n:
        kk = min(1024, n - len(got_ids))
        mi = np.empty((1, kk), np.int32); mv = np.empty((1, kk), np.float64)
        _lib.call("hipts_topk_after", _lib.ptr(dev), ctypes.c_int64(n), kk, ctypes.c_double(got_vals[-1]), ctypes.c_int64(int(got_ids[-1])), _lib.ptr(mi),
                  _lib.ptr(mv), 0, _lib.current_stream_ptr())
        if got_vals[-1] == -np.inf:
            break                                                 # nothing ranks after a -inf entry with a larger index ... except later -inf entries
        got_ids += list(mi[0]); got_vals += list(mv[0])
    finite = int(np.isfinite(v).sum())
    assert got_ids[:finite] == list(order[:finite])
    np.testing.assert_array_equal(np.array(got_vals[:finite]), v[order[:finite]])
    assert all(x == -np.inf for x in got_vals[finite:])


# --------------------------------------------------------------------------------- character features (config[4] rerank)
def test_cfeatures_rerank_cosine():
    from hiptagsearch.cfeatures import CharacterFeatureIndex, cfeatures_rerank
    rng = np.random.default_rng(3)
    n = 5000
    feats = rng.standard_normal((n, 768)).astype(np.float32)
    feats[100:140] = feats[7] + 0.05 * rng.standard_normal((40, 768)).astype(np.float32)      # near-duplicates of image 7
    paths = ["img%05d.png" % i for i in range(n)]
    ci = CharacterFeatureIndex(encoder=lambda x: np.zeros((len(x), 768), np.float32))
    ci.add_features(paths[: n // 2], feats[: n // 2])
    ci.add_features(paths[n // 2:], feats[n // 2:])
    unit = feats / np.linalg.norm(feats, axis=1, keepdims=True)
    q = feats[7]
    d = ci.differences(q)
    want = 1.0 - unit @ (q / np.linalg.norm(q))
    np.testing.assert_allclose(d, want, atol=2e-6)
    tags = {p: {"a": True} if i % 2 == 0 else {"a": True, "b": True} for i, p in enumerate(paths)}
    docid = {p: i for i, p in enumerate(paths)}
    top10 = [(7, 0.9)]
    res = cfeatures_rerank(top10, [feats[7]], ci, tags, docid, required_tags=["a"], exclude_tags=["b"], threshold=0.05)
    ids = [i for i, _ in res[1:]]
    assert res[0] == (7, 0.9)
    assert set(ids) == {i for i in list(range(100, 140)) + [7] if i % 2 == 0}
    assert all(res[i][1] >= res[i + 1][1] for i in range(1, len(res) - 1))


# ---------------------------------------------------------------- sharded index (query partitioning)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_query_is_bit_identical_to_unsharded(world):
    """All shards on this one GPU, the collectives replaced by their definition (max over ranks, concatenation)."""
    import torch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from hiptagsearch.shard import ShardedSearchEngine, global_bm25_stats, merge_topk
    V, D, K, k = 600, 4001, 300, 100
    ptr, terms = synth.tag_corpus(D, V, seed=11)
    rows = synth.index_vectors(D, K, seed=12)
    qs = [dict(q) for q in synth.queries(40, V, seed=13)]
    qv = np.random.default_rng(14).standard_normal((len(qs), K)).astype(np.float32)
    stats = global_bm25_stats(ptr, terms, V)
    shards = [ShardedSearchEngine(ptr, terms, V, rows, r, world, stats=stats) for r in range(world)]
    local = [s.local_scores(qs, qv) for s in shards]
    max_a = torch.stack([l[2] for l in local]).max(dim=0).values           # all-reduce(MAX)
    max_b = torch.stack([l[3] for l in local]).max(dim=0).values
    cands = [s.local_topk(l[0], l[1], max_a, max_b, k) for s, l in zip(shards, local)]
    ids, vals = merge_topk([c[0] for c in cands], [c[1] for c in cands], k)   # all-gather + merge
    bm = BM25Index(ptr, terms, V)
    idx = Similarity("whole", None, K, capacity=D)
    idx.add_matrix(rows)
    wi, wv = SearchEngine(None, idx, {}, bm, []).score_topk(qs, qv, k)
    assert np.array_equal(ids, wi.astype(np.int64))
    assert vals.tobytes() == wv.tobytes()
    # the shard handles really carry the global statistics
    assert float(shards[-1].bm25.export()["idf"][5]).hex() == float(bm.export()["idf"][5]).hex()


def test_sharded_query_two_processes_gloo():
    """Two ranks (both on this GPU, gloo) through ShardedSearchEngine.score_topk with real collectives."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tools", "sharded_query_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    assert "identical to the unsharded engine" in outs[0]


def test_submit_collect_pipeline_matches_the_synchronous_call(corpus20k):
    """hipts_search_submit / hipts_search_collect: two batches in flight on the two slots give, batch for batch, the bytes of hipts_search --
    batches of different sizes (the last is ragged), a one-query batch in the middle (it takes the one-query path inside submit), and the
    misuse cases are refused: a slot submitted twice, a slot collected without a submit."""
    import hiptagsearch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    ptr, terms, V = corpus20k
    D, K, TOPK = len(ptr) - 1, 128, 50
    bm = BM25Index(ptr, terms, V)
    idx = Similarity("pipe", None, K, capacity=D)
    idx.add_matrix(synth.index_vectors(D, K, seed=46))
    eng = SearchEngine(None, idx, {}, bm, [])
    qs = [dict(q) for q in synth.queries(300, V, seed=77)]
    qv = np.random.default_rng(6).standard_normal((300, K))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    cuts = [0, 96, 97, 224, 300]                                     # batches of 96, 1, 127, 76 queries
    want = [eng.score_topk(qs[a:b], qv[a:b], TOPK) for a, b in zip(cuts[:-1], cuts[1:])]
    got = []
    pending = eng.submit_topk(qs[cuts[0]:cuts[1]], qv[cuts[0]:cuts[1]], TOPK, slot=0)
    for j in range(1, len(cuts) - 1):
        nxt = eng.submit_topk(qs[cuts[j]:cuts[j + 1]], qv[cuts[j]:cuts[j + 1]], TOPK, slot=j & 1)
        got.append(eng.collect_topk(pending))
        pending = nxt
    got.append(eng.collect_topk(pending))
    for (gi, gv), (wi, wv) in zip(got, want):
        np.testing.assert_array_equal(gi, wi)
        assert gv.tobytes() == wv.tobytes()
    t = eng.submit_topk(qs[:4], qv[:4], TOPK, slot=1)
    with pytest.raises(hiptagsearch.HipTagSearchError):
        eng.submit_topk(qs[:4], qv[:4], TOPK, slot=1)                # the slot is taken
    eng.collect_topk(t)
    with pytest.raises(hiptagsearch.HipTagSearchError):
        eng.collect_topk(t)                                           # nothing submitted any more
