"""GPU parity at the workloads BASELINE.json's configs state (not scaled-down stand-ins):

  configs[0]  tagging.py --dir on 32 synthetic 448x448 RGB images, ViT-B/16 geometry -> tags-wd-tagger.txt
  configs[1]  ViT-B/16 @448, batch 64, exactly as bench.py runs it (two sub-batch streams, deferred join)
  configs[2]  100k tag documents: BM25 + 300-d index product, top-100, 32 queries
  configs[4]  rerank cosine over 100k x 768 feature rows
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anime-illust-image-searcher_amd")
LOGIT_TOL = 1e-3          # north_star: ViT logits within 1e-3 (absolute, float32)


def test_config1_batch64_two_streams_deferred_join_matches_oracle():
    """The configuration bench.py times: batch 64, depth 12, default two sub-batch streams, deferred join with the
    consumer on a side stream, several forwards back to back.  Images 0, 31, 32, 63 (first / last of each sub-batch)
    against the float32 oracle; every image bit-equal to its own single-image forward (batch invariance); the forward
    repeated gives identical bits."""
    import torch
    from hiptagsearch import _lib, synth
    from hiptagsearch.tagger import ViTTagger
    from oracle import vit as ovit
    cfg = dict(synth.VIT_B16_448)
    w = synth.vit_weights(cfg, seed=0)
    B = 64
    imgs = synth.images_u8(B, 448, seed=1234)
    model = ViTTagger(cfg, w, max_batch=B)
    dimgs = torch.from_numpy(imgs).cuda()
    _lib.call("hipts_vit_set_deferred_join", model._h, 1)
    side = torch.cuda.Stream()
    outs = [torch.empty((B, cfg["num_classes"]), dtype=torch.float32, device="cuda") for _ in range(3)]
    pouts = [torch.empty_like(o) for o in outs]
    copies = []
    for lg, pr in zip(outs, pouts):                                   # three forwards in flight, as in the bench loop
        model.forward_u8(dimgs, logits=lg, probs=pr)
        with torch.cuda.stream(side):
            _lib.call("hipts_vit_join", model._h, _lib.current_stream_ptr())
            copies.append((lg.clone(), pr.clone()))
    torch.cuda.synchronize()
    _lib.call("hipts_vit_set_deferred_join", model._h, 0)
    got = copies[0][0].cpu().numpy()
    for lg, pr in copies[1:]:
        np.testing.assert_array_equal(lg.cpu().numpy(), got)
    probs = copies[2][1].cpu().numpy()
    np.testing.assert_allclose(probs, 1 / (1 + np.exp(-got.astype(np.float64))), atol=2e-7)
    pick = [0, 31, 32, 63]
    x = ovit.preprocess_u8_nhwc(imgs[pick])
    want = ovit.vit_forward(ovit.to_torch(w), x, patch=cfg["patch"], heads=cfg["heads"], eps=cfg["ln_eps"], gelu_kind="tanh").numpy()
    err = np.abs(got[pick] - want).max()
    print("batch-64 two-stream forward: max |logit error| on images 0/31/32/63 = %.3e" % err)
    assert err <= LOGIT_TOL
    _lib.call("hipts_vit_set_sub_batches", model._h, 1)
    for i in range(B):                                                # batch invariance: the same instruction sequence per image
        one, _ = model.forward_u8(imgs[i:i + 1])
        assert np.array_equal(one[0], got[i]), "image %d differs from its single-image forward" % i
    _lib.call("hipts_vit_set_sub_batches", model._h, 0)


def test_deferred_join_survives_a_change_of_batch_size():
    """workspaces are carved by image offset, so an unjoined forward of 32 images (sub-batch 1 = images 16..31)
    followed by one of 64 (sub-batch 0 = images 0..31) must be ordered inside the library; likewise host-staged input."""
    import torch
    from hiptagsearch import _lib, synth
    from hiptagsearch.tagger import ViTTagger
    cfg = dict(synth.VIT_B16_448)
    cfg["depth"] = 3                                                  # real token geometry, a few layers: long enough kernels to overlap
    w = synth.vit_weights(cfg, seed=5)
    model = ViTTagger(cfg, w, max_batch=64)
    a = synth.images_u8(32, 448, seed=31)
    b = synth.images_u8(64, 448, seed=32)
    c = synth.images_u8(16, 448, seed=33)
    want = {}
    for name, im in (("a", a), ("b", b), ("c", c)):
        want[name], _ = model.forward_u8(im)                          # joined, host in/out
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    oa = torch.empty((32, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    ob = torch.empty((64, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    oc = torch.empty((16, cfg["num_classes"]), dtype=torch.float32, device="cuda")
    _lib.call("hipts_vit_set_deferred_join", model._h, 1)
    for _ in range(3):
        model.forward_u8(da, logits=oa, want="logits")                # 32 images, left unjoined
        model.forward_u8(db, logits=ob, want="logits")                # 64 images: other image ranges per stream
        model.forward_u8(c, logits=oc, want="logits")                 # host input staged through the shared buffer
        _lib.call("hipts_vit_join", model._h, _lib.current_stream_ptr())
        torch.cuda.synchronize()
        np.testing.assert_array_equal(ob.cpu().numpy(), want["b"])
        np.testing.assert_array_equal(oc.cpu().numpy(), want["c"])
    _lib.call("hipts_vit_set_deferred_join", model._h, 0)


@pytest.fixture(scope="module")
def corpus100k():
    from hiptagsearch import synth
    D, V = 100_000, 10_000
    ptr, terms = synth.tag_corpus(D, V, seed=42)                      # the bench corpus
    return ptr, terms, V


def test_config2_100k_docs_top100_rank_equal(corpus100k):
    """configs[2] at its stated size: 100k documents, 300-d index, top-100, 32 queries -- BM25 and combined scores
    byte-equal, ids equal (webui.py:345-383,191-192 restated by the oracle)."""
    import torch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25
    from oracle import search as osearch
    ptr, terms, V = corpus100k
    D, K, NQ, TOPK = len(ptr) - 1, 300, 32, 100
    rows = synth.index_vectors(D, K, seed=46)
    bm = BM25Index(ptr, terms, V)
    idx = Similarity("cfg2", None, K, capacity=D)
    idx.add_matrix(rows)
    eng = SearchEngine(None, idx, {}, bm, [])
    qs = synth.queries(NQ, V, seed=43)
    qv = np.random.default_rng(5).standard_normal((NQ, K))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    # the oracle's own statistics from the token lists (genmodel.py:51-99 restated), compared with what the device built
    docs = [[str(t) for t in terms[ptr[d]:ptr[d + 1]]] for d in range(D)]
    corpus, idf, avgdl, _, dl = obm25.bm25_build(docs, {str(i): i for i in range(V)})
    optr, oterm, otf = obm25.to_csr(corpus)
    idf_arr = np.zeros(V)
    for t, v in idf.items():
        idf_arr[t] = v
    e = bm.export()
    np.testing.assert_array_equal(e["csr_ptr"], optr)
    np.testing.assert_array_equal(e["csr_term"], oterm)
    np.testing.assert_array_equal(e["csr_tf"], otf)
    assert e["idf"].tobytes() == idf_arr.tobytes() and float(bm.avgdl).hex() == float(avgdl).hex()
    e = {"csr_ptr": optr, "csr_term": oterm, "csr_tf": otf, "idf": idf_arr, "doc_len": dl}
    got_bm = bm.score([dict(q) for q in qs])
    final_dev = torch.empty((NQ, D), dtype=torch.float64, device="cuda")
    ids, vals = eng.score_topk([dict(q) for q in qs], qv, TOPK, final_out=final_dev)
    final = final_dev.cpu().numpy()
    single_ids = [eng.score_topk([dict(q)], qv[i:i + 1], TOPK)[0][0] for i, q in enumerate(qs[:4])]      # the one-query path
    for i, q in enumerate(qs):
        b = obm25.bm25_score_csr(e["csr_ptr"], e["csr_term"], e["csr_tf"], e["idf"], avgdl, e["doc_len"], [t for t, _ in q], [w for _, w in q])
        assert got_bm[i].tobytes() == b.tobytes(), "query %d BM25" % i
        f = osearch.combine(b, osearch.similarity(rows, qv[i]))
        assert final[i].tobytes() == f.tobytes(), "query %d combined" % i
        wi, wv = osearch.topk(f, TOPK)
        np.testing.assert_array_equal(ids[i], wi)
        assert vals[i].tobytes() == wv.tobytes()
        if i < 4:
            np.testing.assert_array_equal(single_ids[i], wi)


def test_config2_benched_shape_256_queries_fused_topk(corpus100k):
    """The shape bench.py times -- 256 queries x 100k documents through ONE call without `final_out`: the one-pass index product
    (sim_mfma_wide_kernel<8>) and the top-k that combines the scores where it reads them (TopkFused, no stored rows) -- checked
    DIRECTLY against the oracle (webui.py:345-383,191-192 restated), not through the stored-row form: ids and values of the top 100
    for 12 of the queries spread over the batch (the first, the last, one per query block of 32 and three more), byte-equal;
    and the whole batch against the stored-row form of the same call."""
    import torch
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from oracle import bm25 as obm25
    from oracle import search as osearch
    ptr, terms, V = corpus100k
    D, K, NQ, TOPK = len(ptr) - 1, 300, 256, 100
    rows = synth.index_vectors(D, K, seed=46)
    bm = BM25Index(ptr, terms, V)
    idx = Similarity("cfg2b", None, K, capacity=D)
    idx.add_matrix(rows)
    eng = SearchEngine(None, idx, {}, bm, [])
    qs = synth.queries(NQ, V, seed=143)
    qv = np.random.default_rng(15).standard_normal((NQ, K))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    ids, vals = eng.score_topk([dict(q) for q in qs], qv, TOPK)                     # the benched call: no final_out
    assert ids.shape == (NQ, TOPK)
    e = bm.export()
    check = [0, 31, 32, 64, 77, 96, 128, 160, 191, 192, 224, 255]
    for i in check:
        q = qs[i]
        b = obm25.bm25_score_csr(e["csr_ptr"], e["csr_term"], e["csr_tf"], e["idf"], bm.avgdl, e["doc_len"], [t for t, _ in q], [w for _, w in q])
        f = osearch.combine(b, osearch.similarity(rows, qv[i]))
        wi, wv = osearch.topk(f, TOPK)
        np.testing.assert_array_equal(ids[i], wi, err_msg="query %d ids" % i)
        assert vals[i].tobytes() == wv.tobytes(), "query %d values" % i
    final_dev = torch.empty((NQ, D), dtype=torch.float64, device="cuda")
    ids2, vals2 = eng.score_topk([dict(q) for q in qs], qv, TOPK, final_out=final_dev)     # stored rows: combine_kernel + top-k over them
    np.testing.assert_array_equal(ids, ids2)
    assert vals.tobytes() == vals2.tobytes()


def test_config4_rerank_cosine_100k_x_768():
    """configs[4] rerank at its stated size: 100k unit feature rows x 768, the difference 1 - cosine against the k-ordered
    float32 chain of the oracle (orc_sim_chain), bit for bit, single query and a batch."""
    from hiptagsearch.cfeatures import CharacterFeatureIndex
    from oracle import search as osearch
    rng = np.random.default_rng(45)
    D = 100_000
    feats = rng.standard_normal((D, 768)).astype(np.float32)
    ci = CharacterFeatureIndex(encoder=lambda x: np.zeros((len(x), 768), np.float32))
    for s in range(0, D, 25_000):
        ci.add_features(["img%06d.png" % i for i in range(s, s + 25_000)], feats[s:s + 25_000])
    unit = ci.index.matrix()                                           # the rows as stored (unit-normalised on add)
    np.testing.assert_allclose(np.linalg.norm(unit[::997].astype(np.float64), axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(unit[::997], feats[::997] / np.linalg.norm(feats[::997], axis=1, keepdims=True), atol=1e-7)
    for qi in (7, 31337):
        q = feats[qi]
        d = ci.differences(q)
        qn = q / np.float32(np.sqrt(np.sum(q * q)))
        want = np.float32(1.0) - osearch.similarity(unit, qn)
        assert d.tobytes() == want.tobytes()
        assert int(np.argmin(d)) == qi and abs(d[qi]) < 1e-6
    qs = unit[[3, 50_000, 99_999]]
    got = ci.index.query(qs)
    for j in range(3):
        assert got[j].tobytes() == osearch.similarity(unit, qs[j]).tobytes()


def test_config0_tagging_cli_on_32_synthetic_448_images(tmp_path):
    """configs[0]: `python tagging.py --dir D` over 32 synthetic 448x448 RGB PNGs with the contract model's geometry
    (ViT-B/16 @448, synthetic weights): 32 lines, each the selection the oracle makes from the device probabilities of
    that image; --compat reproduces the reference's dropped tail batch: 30 lines (tagging.py:304-338, golden g6)."""
    from PIL import Image
    from hiptagsearch import synth
    from hiptagsearch.tagger import ViTTagger
    from oracle import tags as otags
    os.makedirs(tmp_path / "imgs")
    imgs = synth.images_u8(32, 448, seed=1234)                        # rng(1234) u8 NHWC
    for i, im in enumerate(imgs):
        Image.fromarray(im).save(tmp_path / "imgs" / ("%02d.png" % i))
    cli = os.path.join(PKG, "tagging.py")
    r = subprocess.run([sys.executable, cli, "--dir", "imgs"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read().splitlines()
    assert len(lines) == 32
    by_path = {l.split(",")[0]: l for l in lines}
    assert sorted(by_path) == sorted(os.path.join("imgs", "%02d.png" % i) for i in range(32))
    cfg = dict(synth.VIT_B16_448)
    model = ViTTagger(cfg, synth.vit_weights(cfg, seed=0, trained_like=True), max_batch=32)      # the CLI's stand-in checkpoint (Predictor.load_model)
    _, probs = model.forward_u8(imgs)
    names, cat = synth.label_table(cfg["num_classes"])
    want = otags.predict_lines(probs, names, cat)                      # tagging.py:185-227 restated (pinned by g4)
    assert all(10 <= len(w.split(",")) <= 60 for w in want)            # tens of labels per image, as a trained tagger selects
    for i in range(32):
        p = os.path.join("imgs", "%02d.png" % i)
        assert by_path[p] == p + "," + want[i], "image %d" % i
    os.remove(tmp_path / "tags-wd-tagger.txt")
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--compat"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read().splitlines()) == 30
    # --precise (operand_f16 |= 16: the attention output as a hi | lo pair): the lines are the oracle's selection from THAT mode's probabilities
    os.remove(tmp_path / "tags-wd-tagger.txt")
    r = subprocess.run([sys.executable, cli, "--dir", "imgs", "--precise"], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = open(tmp_path / "tags-wd-tagger.txt", encoding="utf-8").read().splitlines()
    model_p = ViTTagger(dict(cfg, operand_f16=1 | 16), synth.vit_weights(cfg, seed=0, trained_like=True), max_batch=32)
    _, probs_p = model_p.forward_u8(imgs)
    assert not np.array_equal(probs_p, probs)                           # the switch reached the device
    want_p = otags.predict_lines(probs_p, names, cat)
    by_path = {l.split(",")[0]: l for l in lines}
    for i in range(32):
        p = os.path.join("imgs", "%02d.png" % i)
        assert by_path[p] == p + "," + want_p[i], "image %d (--precise)" % i
