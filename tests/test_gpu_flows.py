"""GPU: the flows around the scoring kernels that round 1 left unexercised (A17, f3, loaders).

  character-oriented search mode                      webui.py:255-342,386-388
  tagging.py --after, genmodel.py --update            tagging.py:281-291, genmodel.py:123-148
  feature-index revisions                             gen_cfeatures.py:317-370
  safetensors checkpoint loaders                      tagging.py:146-148 (timm key layout)
"""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anime-illust-image-searcher_amd")


def test_character_oriented_search_mode_end_to_end(tmp_path):
    """find_similar_documents in 'character oriented' mode: top-10 of the combined score -> their images re-encoded by the
    CCIP encoder -> mean feature -> difference to every row of the feature index -> threshold + required / excluded tags
    (webui.py:255-342), against oracle.search.cfeatures_rerank driven by the float32 CPU oracle of the same encoder."""
    from PIL import Image
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.cfeatures import CCIPEncoder, CharacterFeatureIndex, gen_image_ndarray
    from hiptagsearch.d2v import Doc2VecInference
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25, ccip as occip, search as osearch
    V, D, dim, epochs = 60, 40, 300, 5
    rng = np.random.default_rng(9)
    toks = synth.vocab_tokens(V)
    os.makedirs(tmp_path / "imgs")
    # 40 images in 8 "characters": images of one character are noisy copies of one base picture
    base = rng.integers(0, 256, (8, 64, 64, 3), dtype=np.uint8)
    paths, docs = [], []
    for i in range(D):
        ch = i % 8
        im = np.clip(base[ch].astype(np.int32) + rng.integers(-6, 7, base[ch].shape), 0, 255).astype(np.uint8)
        p = str(tmp_path / "imgs" / ("%03d.png" % i))
        Image.fromarray(im).save(p)
        paths.append(p)
        doc = [toks[ch], toks[8 + (i % 5)], toks[20 + (i % 3)], toks[30 + rng.integers(0, 30)]]     # >= 3 tags
        docs.append(list(dict.fromkeys(doc)))
    lines = [p + "," + ",".join(d) for p, d in zip(paths, docs)]
    token2id = {t: i for i, t in enumerate(toks)}
    ptr = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    terms = np.array([token2id[t] for d in docs for t in d], dtype=np.int32)
    m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=dim, seed=44)
    model = Doc2VecInference(m["syn1neg"], m["cum_table"], m["sample_int"], token2id, epochs=epochs)
    rows = model.infer_vectors(docs)
    index = Similarity("idx", None, dim, capacity=D)
    index.add_matrix(rows)
    bm = BM25Index.from_tokens(docs, token2id)
    # the feature index, built as gen_cfeatures.py does (encoder over every image)
    ccfg = dict(synth.CCIP_TINY)
    cw = synth.ccip_weights(ccfg, seed=3)
    enc = CCIPEncoder(ccfg, cw, max_batch=8)
    cindex = CharacterFeatureIndex(enc)
    if enc.out_dim != 768:
        cindex.index = Similarity("c", None, enc.out_dim)
    arrs = [gen_image_ndarray(p, ccfg["image_size"]) for p in paths]
    feats = np.concatenate([cindex.ccip_batch_extract_features(arrs[s:s + 8]) for s in range(0, D, 8)])
    cindex.add_features(paths, feats)
    unit = cindex.index.matrix()
    eng = SearchEngine(model, index, token2id, bm, lines, search_mode="character oriented")
    eng.cindex = cindex
    # oracle side: CPU float32 encoder of the same graph for the query-time features
    tw = occip.to_torch(cw)

    def oracle_feature(path):
        a = gen_image_ndarray(path, ccfg["image_size"])
        import torch
        return occip.metaformer_forward(tw, torch.from_numpy(a[None].astype(np.float32)), dims=ccfg["dims"], depths=ccfg["depths"]).numpy()[0]

    corpus, idf, avgdl, _, dl = obm25.bm25_build(docs, token2id)
    # a cut between same-character pairs and different-character pairs of THIS encoder (the reference's constant is the metric
    # model's; see cfeatures.DEFAULT_COSINE_DIFF_THRESHOLD): midpoint of the two populations
    sims = unit @ unit.T
    same = np.array([[i % 8 == j % 8 for j in range(D)] for i in range(D)])
    lo, hi = (1 - sims[same]).max(), (1 - sims[~same]).min()
    print("1 - cosine: same character <= %.4f, different character >= %.4f" % (lo, hi))
    assert lo < hi
    cindex.cosine_diff_threshold = float((lo + hi) / 2)
    from tests_helpers_d2v import oracle_query_vector                     # noqa: E402  (defined below via sys.modules)
    for query, req, exc in [(toks[0], [], []), (toks[1] + " " + toks[9] + ":+1", [toks[9]], []), (toks[2] + " " + toks[21] + ":-1", [], [toks[21]])]:
        got = eng.find_similar_documents(query, topn=50)
        qvec = oracle_query_vector(model, query, dim)
        d2v_terms, allw, bm_terms = osearch.parse_query(query)
        b = obm25.bm25_score(corpus, idf, avgdl, D, dl, osearch.query_weights(bm_terms, token2id))
        final = osearch.combine(b, osearch.similarity(rows, qvec.astype(np.float32)))
        want = osearch.cfeatures_rerank(final, 50, req, exc, lines, paths, unit, oracle_feature, cindex.cosine_diff_threshold)
        assert [d for d, _ in got[:10]] == [d for d, _ in want[:10]], query                     # the pinned top-10 (webui.py:332-333)
        np.testing.assert_array_equal([s for _, s in got[:10]], [s for _, s in want[:10]])
        assert sorted(d for d, _ in got[10:]) == sorted(d for d, _ in want[10:]), query          # who passes threshold + tag filters
        gs, ws = dict(got[10:]), dict(want[10:])
        for d in gs:                                                                              # device bf16 encoder vs float32 oracle encoder
            assert abs(gs[d] - ws[d]) <= 5e-3
        tail = [s for _, s in got[10:]]
        assert all(tail[i] >= tail[i + 1] for i in range(len(tail) - 1))                          # webui.py:330
        for d, _ in got[10:]:
            assert all(t in docs[d] for t in req) and all(t not in docs[d] for t in exc)
        ch = {d % 8 for d, _ in got[10:]}
        assert len(ch) <= 1 or len(got) == 10                                                     # one character's images pass the cut


def _install_helper():
    """oracle-side query vector (webui.py:82-117) from the C oracle's Doc2Vec inference with the product's start vectors / seeds."""
    import types
    mod = types.ModuleType("tests_helpers_d2v")

    def oracle_query_vector(model, query, dim):
        from hiptagsearch.d2v import pseudorandom_weak_vector
        from oracle import d2v as od2v, search as osearch
        d2v_terms, allw, _ = osearch.parse_query(query)

        def infer(words):
            p = np.array([0, len(words)], dtype=np.int64)
            ids = np.array([model.key_to_index.get(t, -1) for t in words], dtype=np.int32)
            v0 = pseudorandom_weak_vector(dim, " ".join(words))[None]
            seeds = np.asarray([model._seed_for(words)], dtype=np.uint64)
            return od2v.infer(model.syn1neg, model.cum_table, model.sample_int, p, ids, v0, seeds, model.epochs)[0]
        return osearch.query_vector(d2v_terms, allw, infer, dim)
    mod.oracle_query_vector = oracle_query_vector
    sys.modules["tests_helpers_d2v"] = mod


_install_helper()


def _png(path, rng, shape=(48, 64, 3)):
    from PIL import Image
    Image.fromarray(rng.integers(0, 256, shape, dtype=np.uint8)).save(path)


def test_tagging_after_and_genmodel_update(tmp_path):
    """tagging.py --after DATE appends only files changed on or after DATE and keeps a .bak (tagging.py:281-291);
    genmodel.py --update infers only the new tail, rebuilds BM25 over everything with the OLD dictionary
    (genmodel.py:123-148,134,177) and the query function sees the new documents."""
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / "imgs")
    for i in range(12):
        _png(str(tmp_path / "imgs" / ("a%02d.png" % i)), rng)
    old = time.time() - 10 * 86400
    for i in range(12):
        os.utime(tmp_path / "imgs" / ("a%02d.png" % i), (old, old))
    run = lambda script, *a: subprocess.run([sys.executable, os.path.join(PKG, script)] + list(a), cwd=tmp_path, capture_output=True, text=True, timeout=600)
    # --after without an existing tag file: the reference prints and exits 1 (tagging.py:289-291)
    r = run("tagging.py", "--dir", "imgs", "--model", "vit-tiny", "--after", "2000-01-01")
    assert r.returncode == 1 and "tags-wd-tagger.txt not found" in r.stdout
    r = run("tagging.py", "--dir", "imgs", "--model", "vit-tiny", "--batch", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    first = open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read()
    assert len(first.splitlines()) == 12
    r = run("genmodel.py", "--synthetic-d2v", "--epochs", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    idx_lines = open(tmp_path / "tags-wd-tagger_doc2vec_idx.csv", encoding="utf-8").read().splitlines()
    # new files arrive; ctime filter (tagging.py:266-274 uses st_ctime: a new file qualifies, an old one whose ctime is now does too,
    # so the date is set to tomorrow for the negative case and to today for the positive one)
    for i in range(5):
        _png(str(tmp_path / "imgs" / ("b%02d.png" % i)), rng)
    import datetime
    tomorrow = (datetime.date.today() + datetime.timedelta(days=1)).isoformat()
    r = run("tagging.py", "--dir", "imgs", "--model", "vit-tiny", "--after", tomorrow)
    assert r.returncode == 0 and "0 files found after" in r.stdout
    assert open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read() == first
    assert open(tmp_path / "tags-wd-tagger.txt.bak", encoding="utf-8").read() == first
    # keep only the b* files young: the reference filters on ctime, which utime cannot set -- so count instead
    r = run("tagging.py", "--dir", "imgs", "--model", "vit-tiny", "--after", datetime.date.today().isoformat(), "--batch", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    second = open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read()
    assert second.startswith(first)                                        # appended (tagging.py:293), nothing rewritten
    new_lines = second[len(first):].splitlines()
    assert {os.path.basename(l.split(",")[0]) for l in new_lines} >= {"b%02d.png" % i for i in range(5)}
    # drop re-tagged a* lines (their ctime is recent in this sandbox) so the update sees exactly five new documents
    keep = first + "".join(l + "\n" for l in new_lines if os.path.basename(l.split(",")[0]).startswith("b"))
    open(tmp_path / "tags-wd-tagger.txt", "w", encoding="utf-8").write(keep)
    r = run("genmodel.py", "--update", "--epochs", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    idx2 = open(tmp_path / "tags-wd-tagger_doc2vec_idx.csv", encoding="utf-8").read().splitlines()
    n_new = len(idx2) - len(idx_lines)
    assert idx2[:len(idx_lines)] == idx_lines and n_new >= 0
    assert ("update index: %d files" % n_new) in r.stdout
    assert open(tmp_path / "tags-wd-tagger_doc2vec_idx.csv.bak", encoding="utf-8").read().splitlines() == idx_lines
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        import pickle
        from hiptagsearch import search
        from hiptagsearch.index import Similarity
        assert len(Similarity.load("doc2vec_index")) == len(idx2)
        assert pickle.load(open("bm25_D", "rb")) == len(idx2)
        eng = search.load_engine()
        if n_new:
            tag = idx2[-1].split(",")[1]
            res = eng.find_similar_documents(tag + ":+1", topn=50)
            assert all(tag in idx2[d].split(",")[1:] for d, _ in res) and len(res) >= 1
    finally:
        os.chdir(cwd)


def test_feature_index_revisions(tmp_path):
    """gen_cfeatures.py --after: timestamped backup directory, revision N+1 = copy of revision N + the new rows, the csv
    appended in place; the loader takes the highest revision and tolerates a csv longer than the index (crash between the
    per-batch csv append and the final save)."""
    sys.path.insert(0, PKG)
    from hiptagsearch import cfeatures as cf
    from hiptagsearch.index import Similarity
    rng = np.random.default_rng(5)
    os.makedirs(tmp_path / "imgs")
    for i in range(6):
        _png(str(tmp_path / "imgs" / ("%02d.png" % i)), rng)
    run = lambda *a: subprocess.run([sys.executable, os.path.join(PKG, "gen_cfeatures.py"), "--arch", "tiny", "--batch", "4"] + list(a),
                                    cwd=tmp_path, capture_output=True, text=True, timeout=600)
    r = run("--dir", "imgs", "--after", "2000-01-01")                     # no index yet: the reference fails in max([]) (gen_cfeatures.py:333)
    assert r.returncode != 0 and "ValueError" in r.stderr
    for d in os.listdir(tmp_path):
        if os.path.isdir(tmp_path / d) and d != "imgs":
            os.rmdir(tmp_path / d)                                        # the (empty) backup directory of the failed run
    if os.path.exists(tmp_path / "charactor-featues-idx.csv"):
        os.remove(tmp_path / "charactor-featues-idx.csv")
    r = run("--dir", "imgs")
    assert r.returncode == 0, r.stderr[-2000:]
    rows0 = Similarity.load(str(tmp_path / "charactor-featues-idx")).matrix()
    assert cf.get_current_cfeature_number(str(tmp_path)) == 0 and len(rows0) == 6
    for rev in (1, 2):
        r = run("--dir", "imgs", "--after", "2000-01-01")
        assert r.returncode == 0, r.stderr[-2000:]
        assert cf.get_current_cfeature_number(str(tmp_path)) == rev
        rows = Similarity.load(str(tmp_path / ("charactor-featues-idx%d" % rev))).matrix()
        assert len(rows) == 6 * (rev + 1)
        np.testing.assert_array_equal(rows[:6], rows0)                    # old rows copied unchanged
        np.testing.assert_array_equal(rows[-6:], rows0)                   # the same images encode to the same rows
        time.sleep(1.1)                                                   # backup directories are named to the second
    backups = [d for d in os.listdir(tmp_path) if os.path.isdir(tmp_path / d) and d != "imgs"]
    assert len(backups) == 2 and all(len(b) == 15 and b[8] == "_" for b in backups)
    assert any(os.path.exists(tmp_path / b / "charactor-featues-idx1.npy") for b in backups)      # second run backed up revision 1 too
    np.testing.assert_array_equal(Similarity.load(str(tmp_path / "charactor-featues-idx")).matrix(), rows0)   # revision 0 untouched
    ci = cf.CharacterFeatureIndex.load_latest(lambda x: None, dirpath=str(tmp_path))
    assert len(ci.index) == 18 and len(ci.paths) == 18
    with open(tmp_path / "charactor-featues-idx.csv", "a", encoding="utf-8") as f:
        f.write("imgs/never-encoded.png\n")                               # crash after the csv append
    ci = cf.CharacterFeatureIndex.load_latest(lambda x: None, dirpath=str(tmp_path))
    assert len(ci.paths) == 18 and "never-encoded" not in ci.paths[-1]
    open(tmp_path / "charactor-featues-idx.csv", "w").write("only-one\n")
    with pytest.raises(ValueError):
        cf.CharacterFeatureIndex.load_latest(lambda x: None, dirpath=str(tmp_path))


def test_safetensors_checkpoint_loaders(tmp_path):
    """tagging.py:146-148 / gen_cfeatures.py:112-118 load a checkpoint file; here: timm-layout safetensors."""
    from safetensors.numpy import save_file
    from hiptagsearch import synth
    from hiptagsearch.cfeatures import CCIPEncoder
    from hiptagsearch.tagger import EvaTagger, ViTTagger
    for cls_, cfg, wfn, size in ((ViTTagger, synth.VIT_TINY, synth.vit_weights, 64), (EvaTagger, synth.EVA02_TINY, synth.eva_weights, 56)):
        w = wfn(dict(cfg), seed=4)
        f = str(tmp_path / (cls_.__name__ + ".safetensors"))
        save_file({k: np.ascontiguousarray(v) for k, v in w.items()}, f)
        imgs = synth.images_u8(2, size, seed=6)
        a, _ = cls_(dict(cfg), w, max_batch=2).forward_u8(imgs)
        b, _ = cls_.from_safetensors(f, dict(cfg), max_batch=2).forward_u8(imgs)
        np.testing.assert_array_equal(a, b)
    ccfg = dict(synth.CCIP_TINY)
    cw = synth.ccip_weights(ccfg, seed=3)
    f = str(tmp_path / "ccip.safetensors")
    save_file({k: np.ascontiguousarray(v) for k, v in cw.items()}, f)
    imgs = synth.images_u8(2, ccfg["image_size"], seed=7)
    np.testing.assert_array_equal(CCIPEncoder(ccfg, cw, max_batch=2).forward_u8(imgs), CCIPEncoder.from_safetensors(f, ccfg, max_batch=2).forward_u8(imgs))
    # the CLI takes the file too
    from PIL import Image
    os.makedirs(tmp_path / "imgs")
    for i in range(3):
        Image.fromarray(synth.images_u8(1, 64, seed=20 + i)[0]).save(tmp_path / "imgs" / ("%d.png" % i))
    r = subprocess.run([sys.executable, os.path.join(PKG, "tagging.py"), "--dir", "imgs", "--model", "vit-tiny", "--checkpoint",
                        str(tmp_path / "ViTTagger.safetensors")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(open(tmp_path / "tags-wd-tagger.txt").read().splitlines()) == 3


def test_ccip_metric_and_calibrated_threshold():
    """hipts_ccip_metric (gen_cfeatures.py:257-274's call shape: features [n,768] -> differences [n,n]) in its cosine form, bit-equal to
    the oracle's restatement (unit rows, k-ordered fmaf chain), and the rerank cut DERIVED from labelled features the way the
    reference's constant was (best F1 over same- / different-character pairs) instead of borrowed from the metric model's scale."""
    from hiptagsearch import cfeatures
    from oracle import search as osearch
    rng = np.random.default_rng(21)
    n_cl, per, dim = 6, 8, 768
    centers = rng.standard_normal((n_cl, dim)).astype(np.float32)
    feats = np.concatenate([c + 0.35 * rng.standard_normal((per, dim)).astype(np.float32) for c in centers]) * np.float32(3.7)
    labels = np.repeat(np.arange(n_cl), per)
    got = cfeatures.ccip_batch_differences(feats)
    # oracle: x / sqrt(fmaf-chain sum of squares) in float32, Gram rows by the same chain, 1 - s
    unit = np.empty_like(feats)
    for i, row in enumerate(feats):
        s = osearch.similarity(row[None, :], row)[0]
        unit[i] = row / np.sqrt(s, dtype=np.float32)
    want = np.stack([np.float32(1.0) - osearch.similarity(unit, unit[i]) for i in range(len(unit))])
    assert got.dtype == np.float32 and got.shape == (n_cl * per, n_cl * per)
    np.testing.assert_array_equal(got, want)
    assert np.abs(np.diag(got)).max() < 1e-5 and np.array_equal(got, got.T)
    assert cfeatures.ccip_difference(feats[0], feats[1]) == float(got[0, 1])
    thr, f1 = cfeatures.calibrate_threshold(feats, labels)
    same = labels[:, None] == labels[None, :]
    off = ~np.eye(len(labels), dtype=bool)
    assert f1 == 1.0 and got[same & off].max() < thr < got[~same].min(), (thr, f1)
