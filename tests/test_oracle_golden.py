"""CPU: the oracle restatements are pinned, bit for bit, to the golden vectors captured from the
reference's own numpy code (tests/golden/make_golden.py), and the product's host logic is checked
against the same vectors."""
import json
import os

import numpy as np
import pytest

from oracle import bm25 as obm25
from oracle import search as osearch
from oracle import tags as otags


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def fh(x):
    return float.fromhex(x)


# ---------------------------------------------------------------- G1: BM25 build
@pytest.mark.parametrize("case", ["tiny", "d1000"])
def test_bm25_build_matches_reference(golden_dir, case):
    g = _load(golden_dir, "g1_bm25_build.json")[case]
    corpus, idf, avgdl, D, dl = obm25.bm25_build(g["docs"], g["token2id"])
    assert D == g["D"] and type(D).__name__ == g["D_type"]
    assert [{str(k): v for k, v in d.items()} for d in corpus] == g["corpus"]
    assert [list(d.keys()) for d in corpus] == [[int(k) for k in d.keys()] for d in g["corpus"]]   # dict order too
    assert list(idf.keys()) == g["idf_keys"]
    assert [float(v).hex() for v in idf.values()] == g["idf_hex"]
    assert type(next(iter(idf.values()))).__name__ == g["idf_type"]
    assert float(avgdl).hex() == g["avgdl_hex"] and type(avgdl).__name__ == g["avgdl_type"]
    assert dl.tolist() == g["doc_lengths"] and str(dl.dtype) == g["doc_lengths_dtype"]


# ---------------------------------------------------------------- G2: BM25 score
@pytest.mark.parametrize("case", ["tiny", "d1000"])
def test_bm25_score_matches_reference(golden_dir, case):
    g1 = _load(golden_dir, "g1_bm25_build.json")[case]
    g2 = _load(golden_dir, "g2_bm25_score.json")[case]
    corpus, idf, avgdl, D, dl = obm25.bm25_build(g1["docs"], g1["token2id"])
    ptr, terms, tfs = obm25.to_csr(corpus)
    V = max(max(idf.keys()) + 1, 1)
    idf_arr = np.zeros(V)
    for k, v in idf.items():
        idf_arr[k] = v
    for q, want_hex in zip(g2["queries"], g2["scores_hex"]):
        qw = {int(k): int(v) for k, v in q}
        want = np.array([fh(x) for x in want_hex])
        got = obm25.bm25_score(corpus, idf, avgdl, D, dl, qw)
        assert got.dtype == np.float64
        assert got.tobytes() == want.tobytes()
        got_csr = obm25.bm25_score_csr(ptr, terms, tfs, idf_arr, avgdl, dl, list(qw.keys()), list(qw.values()))
        assert got_csr.tobytes() == want.tobytes()


# ---------------------------------------------------------------- G3: MCut
def test_mcut_matches_reference(golden_dir):
    for c in _load(golden_dir, "g3_mcut.json"):
        p = np.array([fh(x) for x in c["probs_hex"]])
        assert float(otags.mcut_threshold(p)).hex() == c["thresh_hex"]


def test_mcut_known_answer():
    assert otags.mcut_threshold(np.array([.9, .85, .3, .28, .05])) == pytest.approx(0.575)


# ---------------------------------------------------------------- G4: predict post-processing
def test_predict_lines_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g4_predict.npz"))
    g = _load(golden_dir, "g4_predict.json")
    import torch
    probs = torch.sigmoid(torch.from_numpy(z["logits"])).numpy()      # F.sigmoid, tagging.py:176
    lines = otags.predict_lines(probs, g["names"], z["category"])
    assert lines == g["lines"]
    # the numpy sigmoid restatement agrees with torch's to float32 rounding, and yields the same lines here
    np.testing.assert_allclose(otags.sigmoid_f32(z["logits"]), probs, rtol=0, atol=1.2e-7)


# ---------------------------------------------------------------- G5: result filter (oracle and product host logic)
def test_filter_searched_result_matches_reference(golden_dir):
    from hiptagsearch.search import filter_searched_result as product_filter
    for c in _load(golden_dir, "g5_filter.json"):
        inp = [(int(a), fh(b)) for a, b in c["in"]]
        want = [(int(a), fh(b)) for a, b in c["out"]]
        assert osearch.filter_searched_result(inp) == want
        assert product_filter(inp) == want


# ---------------------------------------------------------------- G6: batching loop line counts
def test_compat_line_counts(golden_dir):
    """tagging.py:304-338 never consumes its last submitted batch; the product's --compat batching
    reproduces the reference's line counts, the default mode writes every file."""
    for c in _load(golden_dir, "g6_batching.json"):
        N = c["N"]
        batches = [list(range(i, min(i + 10, N))) for i in range(0, N, 10)]
        assert sum(len(b) for b in batches[:-1]) == c["lines"]
        assert c["batch_sets_match_listing_order"]


# ---------------------------------------------------------------- G8: tag-file parser
def test_read_documents_matches_reference(golden_dir, tmp_path, monkeypatch):
    from hiptagsearch.textio import read_documents_and_gen_idx_text
    g = _load(golden_dir, "g8_read_documents.json")
    monkeypatch.chdir(tmp_path)
    open("tags-wd-tagger.txt", "w", encoding="utf-8").write(g["input"])
    docs, tagged = read_documents_and_gen_idx_text("tags-wd-tagger.txt")
    assert docs == g["docs"]
    assert open("tags-wd-tagger_doc2vec_idx.csv", encoding="utf-8").read() == g["idx_text"]
    assert [t[1] for t in tagged] == [[i] for i in range(len(docs))]


# ---------------------------------------------------------------- G7: image preparation
def test_prepare_image_matches_reference(golden_dir):
    from PIL import Image
    from hiptagsearch.tagger import Predictor
    z = np.load(os.path.join(golden_dir, "g7_prepare_image.npz"))
    P = Predictor.__new__(Predictor)
    for mode in ("RGBA", "LA", "RGB", "L"):
        arr = z["in_" + mode]
        img = Image.fromarray(arr, mode)
        np.testing.assert_array_equal(np.asarray(P.prepare_image(img)), z["out_" + mode])


# ---------------------------------------------------------------- oracle internals
def test_similarity_chain_is_sequential_fma():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((17, 300)).astype(np.float32)
    q = rng.standard_normal(300).astype(np.float32)
    got = osearch.similarity(a, q)
    # float64 emulation of one fused step: exact product, one rounding of the sum (double rounding
    # could differ in rare cases, so compare with a tolerance of 1 ulp and require most to be exact)
    acc = np.zeros(17, dtype=np.float32)
    for k in range(300):
        acc = (a[:, k].astype(np.float64) * np.float64(q[k]) + acc.astype(np.float64)).astype(np.float32)
    assert np.mean(got == acc) > 0.9
    np.testing.assert_allclose(got, acc, rtol=3e-7)


def test_stable_rank_ties_and_inf():
    s = np.array([0.5, -np.inf, 0.7, 0.5, 0.7, -np.inf])
    assert osearch.stable_rank(s).tolist() == [2, 4, 0, 3, 1, 5]
    py = [i for i, _ in sorted(enumerate(s), key=lambda it: -it[1])]      # webui.py:191-192
    assert osearch.stable_rank(s).tolist() == py


def test_query_parsers():
    d2v, allw, bm = osearch.parse_query("1girl blue_eyes:+2 hat:-3 foo:bar a_(b):2")
    assert d2v == [("1girl", 1), ("blue_eyes", 2), ("hat", -3), ("foo:bar", 1), ("a_\\(b\\)", 2)]
    assert allw == 3
    assert bm == [("1girl", "plain", 1), ("blue_eyes", "require", 2), ("hat", "exclude", -3), ("foo:bar", "plain", 1),
                  ("a_(b)", "exclude", 2)]


# ---------------------------------------------------------------- G9: CCIP preprocessing
def test_ccip_preprocess_matches_reference(golden_dir):
    import hashlib
    from PIL import Image
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "cfeatures_host", os.path.join(os.path.dirname(golden_dir), "..", "anime-illust-image-searcher_amd", "hiptagsearch", "cfeatures.py"))
    src = open(spec.origin).read()
    # host-only functions: exec the module text without its package-relative import
    ns = {}
    exec(compile(src.replace("from .index import Similarity", "Similarity = None"), spec.origin, "exec"), ns)
    for c in _load(golden_dir, "g9_ccip_preprocess.json"):
        arr = np.random.default_rng(c["seed"]).integers(0, 256, (c["h"], c["w"], 3), dtype=np.uint8)
        o = ns["_preprocess_image"](Image.fromarray(arr), size=384)
        assert str(o.dtype) == c["dtype"] and list(o.shape) == c["shape"]
        assert [float(v).hex() for v in o[:, ::96, ::96].ravel()] == c["sample_hex"]
        assert hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest() == c["sha256"]
