"""GPU end-to-end: the CLI surface of the reference (tagging.py --dir, genmodel.py, query function)
on a small synthetic image directory, through the device kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anime-illust-image-searcher_amd")


def test_tagging_genmodel_query_pipeline(tmp_path, monkeypatch):
    from PIL import Image
    from hiptagsearch import search, synth
    from hiptagsearch.tagger import Predictor
    monkeypatch.chdir(tmp_path)
    os.makedirs("imgs/sub")
    rng = np.random.default_rng(0)
    n = 23
    for i in range(n):
        arr = rng.integers(0, 256, (40 + i, 64, 3), dtype=np.uint8)
        Image.fromarray(arr).save("imgs/%s%03d.png" % ("sub/" if i % 3 == 0 else "", i))
    open("imgs/notes.txt", "w").write("not an image")
    # tagging stage with the tiny config (same code path as ViT-B/16, seconds instead of minutes of init)
    pred = Predictor(max_batch=8)
    pred.load_model(cfg=synth.VIT_TINY, seed=3)
    pred.process_directory("imgs", batch_size=8)
    lines = open("tags-wd-tagger.txt", encoding="utf-8").read().splitlines()
    assert len(lines) == n                                         # default mode writes every file (no tail drop)
    assert sorted(l.split(",")[0] for l in lines) == sorted(pred.list_files_recursive("imgs"))
    assert all(len(l.split(",")) >= 2 for l in lines)
    # the multi-process decode pool and the pre-decoded shards give the same lines (same uint8 images, same device path)
    from hiptagsearch import pipeline
    for kw in ({"workers": 2}, {"shards": "shards"}):
        if "shards" in kw:
            assert pipeline.write_shards(pred.list_files_recursive("imgs"), "shards", size=pred.cfg["image_size"], workers=2, per_shard=10) == n
        os.remove("tags-wd-tagger.txt")
        pred.process_directory("imgs", batch_size=8, **kw)
        assert open("tags-wd-tagger.txt", encoding="utf-8").read().splitlines() == lines
    # compat mode reproduces the reference's dropped tail batch: (ceil(23/10)-1)*10 = 20 lines
    os.remove("tags-wd-tagger.txt")
    pred2 = Predictor(max_batch=8, compat=True)
    pred2.tagger_model, pred2.selector, pred2.tag_names, pred2.cfg = pred.tagger_model, pred.selector, pred.tag_names, pred.cfg
    pred2.process_directory("imgs", batch_size=10)
    assert len(open("tags-wd-tagger.txt").read().splitlines()) == 20
    # index stage
    os.remove("tags-wd-tagger.txt")
    open("tags-wd-tagger.txt", "w").write("\n".join(lines) + "\n")
    r = subprocess.run([sys.executable, os.path.join(PKG, "genmodel.py"), "--epochs", "5"],            # trains the Doc2Vec model (genmodel.py:159-162)
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in ("doc2vec_model", "doc2vec_index", "doc2vec_dictionary", "bm25_corpus", "bm25_idf", "bm25_avgdl", "bm25_D", "bm25_doc_lengths",
              "tags-wd-tagger_doc2vec_idx.csv"):
        assert os.path.exists(f), f
    # query stage
    eng = search.load_engine()
    search.set_engine(eng)
    docs = [l.split(",")[1:] for l in open("tags-wd-tagger_doc2vec_idx.csv").read().splitlines()]
    tag = docs[0][0]
    res = search.find_similar_documents(tag, topn=5)
    assert 1 <= len(res) <= 5 and all(0 <= d < len(docs) for d, _ in res)
    assert res[0][1] == pytest.approx(1.0)
    req = search.find_similar_documents(tag + ":+1", topn=50)
    assert all(tag in docs[d] for d, _ in req)                     # required tag really required
    with pytest.raises(KeyError):
        search.find_similar_documents("definitely_not_a_tag", topn=5)   # webui.py:371 behaviour


def test_gen_cfeatures_cli_builds_and_extends_the_feature_index(tmp_path, monkeypatch):
    """gen_cfeatures.py --dir D [--after DATE]: paths csv + unit-normalised feature index, then an --after append."""
    from PIL import Image
    from hiptagsearch.index import Similarity
    monkeypatch.chdir(tmp_path)
    os.makedirs("imgs/sub")
    rng = np.random.default_rng(1)
    for i in range(7):
        Image.fromarray(rng.integers(0, 256, (50 + i, 70, 3), dtype=np.uint8)).save("imgs/%s%02d.png" % ("sub/" if i % 2 else "", i))
    open("imgs/broken.png", "wb").write(b"not a png")                      # failed loads are skipped, not fatal
    cli = os.path.join(PKG, "gen_cfeatures.py")
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--batch", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    paths = open("charactor-featues-idx.csv", encoding="utf-8").read().splitlines()
    assert len(paths) == 7 and all(p.endswith(".png") and "broken" not in p for p in paths)
    idx = Similarity.load("charactor-featues-idx")
    assert len(idx) == 7 and idx.num_features == 768
    m = idx.matrix()
    np.testing.assert_allclose(np.linalg.norm(m, axis=1), 1.0, atol=1e-5)  # rows are unit vectors (gensim normalises on add)
    sims = idx.query(m[3])[0]
    assert int(np.argmax(sims)) == 3 and sims[3] == pytest.approx(1.0, abs=1e-5)
    # the same image encodes to the same row whatever batch it was in
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--after", "2000-01-01", "--batch", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # --after writes a NEW revision (gen_cfeatures.py:354-370): charactor-featues-idx stays, charactor-featues-idx1 = old rows + new rows
    assert len(Similarity.load("charactor-featues-idx")) == 7
    idx2 = Similarity.load("charactor-featues-idx1")
    assert len(idx2) == 14
    m2 = idx2.matrix()
    p2 = open("charactor-featues-idx.csv", encoding="utf-8").read().splitlines()
    assert p2[:7] == paths
    for i, p in enumerate(paths):
        j = 7 + p2[7:].index(p)
        np.testing.assert_allclose(m2[j], m[i], atol=1e-6)
    # --workers: multi-process decode to uint8, normalisation on the device -- the same features again
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--after", "2000-01-01", "--batch", "4", "--workers", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m3 = Similarity.load("charactor-featues-idx2").matrix()
    p3 = open("charactor-featues-idx.csv", encoding="utf-8").read().splitlines()
    assert len(p3) == 21
    for i, p in enumerate(paths):
        np.testing.assert_allclose(m3[14 + p3[14:].index(p)], m[i], atol=1e-5)
    # --gpu-resize: the threads only decode, the bilinear resize is the device kernel (Pillow's result bit for bit) -- the same features
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--after", "2000-01-01", "--batch", "4", "--gpu-resize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m4 = Similarity.load("charactor-featues-idx3").matrix()
    p4 = open("charactor-featues-idx.csv", encoding="utf-8").read().splitlines()
    assert len(p4) == 28
    for i, p in enumerate(paths):
        np.testing.assert_allclose(m4[21 + p4[21:].index(p)], m[i], atol=1e-5)
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--after", "not-a-date"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid date format" in r.stdout


def test_clis_with_gpu_jpeg_write_the_same_files(tmp_path, monkeypatch):
    """tagging.py / gen_cfeatures.py with --workers N --gpu-jpeg (entropy decoding in the workers, the rest of the JPEG decode on the
    device): the tag file and the feature rows of the plain runs, over baseline and progressive JPEGs of three samplings, a greyscale
    one, a PNG and a file that is not an image."""
    from PIL import Image
    from hiptagsearch.index import Similarity
    monkeypatch.chdir(tmp_path)
    os.makedirs("imgs/sub")
    rng = np.random.default_rng(5)
    for i in range(14):
        h, w = int(rng.integers(40, 400)), int(rng.integers(40, 500))
        small = Image.fromarray(rng.integers(0, 256, (max(2, h // 16), max(2, w // 16), 3), dtype=np.uint8)).resize((w, h), Image.BICUBIC)
        if i == 5:
            small = small.convert("L")
        kw = {} if i == 5 else {"subsampling": i % 3}
        small.save("imgs/%s%02d.jpg" % ("sub/" if i % 2 else "", i), quality=70 + i, progressive=(i % 4 == 1), **kw)
    Image.fromarray(rng.integers(0, 256, (60, 80, 4), dtype=np.uint8), "RGBA").save("imgs/alpha.png")
    open("imgs/broken.jpg", "wb").write(b"\xff\xd8 not a jpeg")
    tag_cli, feat_cli = os.path.join(PKG, "tagging.py"), os.path.join(PKG, "gen_cfeatures.py")
    outs = []
    for extra in ([], ["--workers", "2", "--gpu-resize"], ["--workers", "2", "--gpu-resize", "--gpu-jpeg"]):
        if os.path.exists("tags-wd-tagger.txt"):
            os.remove("tags-wd-tagger.txt")
        r = subprocess.run([sys.executable, tag_cli, "--dir", "imgs", "--batch", "8", "--model", "vit-tiny"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(sorted(open("tags-wd-tagger.txt", encoding="utf-8").read().splitlines()))
    assert len(outs[0]) == 15 and outs[0] == outs[1] == outs[2]
    feats = []
    for k, extra in enumerate(([], ["--workers", "2", "--gpu-jpeg"])):
        for f in os.listdir("."):
            if f.startswith("charactor-featues-idx"):
                os.remove(f)
        r = subprocess.run([sys.executable, feat_cli, "--dir", "imgs", "--batch", "8", "--arch", "tiny"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        paths = open("charactor-featues-idx.csv", encoding="utf-8").read().splitlines()
        m = Similarity.load("charactor-featues-idx").matrix()
        feats.append({p: m[i] for i, p in enumerate(paths)})
    assert sorted(feats[0]) == sorted(feats[1]) and len(feats[0]) == 15
    for p in feats[0]:
        np.testing.assert_allclose(feats[1][p], feats[0][p], atol=1e-5)
