"""GPU: Doc2Vec PV-DBOW training (hipts_d2v_train; genmodel.py:159-162).

  sequential mode  bit-identical to the CPU oracle (orc_d2v_train): the reference's workers=1 order on one wavefront
  parallel mode    all documents of an epoch concurrently, lock-free hidden-layer updates: judged downstream -- nearest
                   neighbours of the document vectors it infers agree with the sequentially trained model's and with the topics
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anime-illust-image-searcher_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("dim,negative,sample", [(300, 5, True), (100, 5, False), (64, 0, True), (40, 3, True)])
def test_sequential_training_is_bit_identical_to_the_oracle(dim, negative, sample):
    from hiptagsearch.d2v import Doc2Vec
    from oracle import d2v as od2v
    from test_oracle_d2v_train import _csr, _topic_corpus
    docs, _ = _topic_corpus(90, topics=3, words_per_topic=10, doc_len=7, seed=dim)
    docs[5] = docs[5] + ["never_seen_%d" % i for i in range(3)]            # (all words are in the vocabulary of build_vocab; OOV only at inference)
    docs[11] = []                                                           # an empty document trains nothing and must not disturb the schedule
    m = Doc2Vec(vector_size=dim, window=50, min_count=1, workers=1, dm=0, negative=negative, sample=1e-2 if sample else 0, seed=7, batch_words=40)
    m.build_vocab(docs)
    k2i, cnt, cum, si = od2v.build_vocab(docs, sample=1e-2)
    assert m.key_to_index == k2i
    ptr, ids = _csr(docs, k2i)
    syn, dv = np.zeros((len(k2i), dim), np.float32), od2v.init_doc_vectors(len(docs), dim, seed=7)
    od2v.train(syn, dv, cum, si if sample else None, ptr, ids, epochs=4, negative=negative, seed=7, batch_words=40)
    m.train(docs, total_examples=m.corpus_count, epochs=4)
    assert m.last_mode == "sequential"
    assert m.syn1neg.tobytes() == syn.tobytes(), "hidden layer differs from the oracle"
    assert m.doc_vectors.tobytes() == dv.tobytes(), "document vectors differ from the oracle"
    assert np.abs(m.syn1neg).max() > 0
    # the trained model infers (genmodel.py:168-169) exactly what the oracle infers from the oracle-trained weights
    from hiptagsearch.d2v import pseudorandom_weak_vector
    inf = m.inference()
    q = [docs[0], docs[1] + ["not_in_vocab"], docs[2][:3]]
    got = inf.infer_vectors(q)
    p2, i2 = _csr(q, k2i)
    v0 = np.stack([pseudorandom_weak_vector(dim, " ".join(d)) for d in q])
    seeds = np.asarray([inf._seed_for(d) for d in q], dtype=np.uint64)
    want = od2v.infer(syn, cum, si if sample else None, p2, i2, v0, seeds, 4, negative=negative)
    assert got.tobytes() == want.tobytes()


def test_parallel_training_quality_matches_sequential_downstream():
    """2 400 documents of 8 topics.  The parallel (throughput) schedule is not reproducible bit for bit; what matters downstream is
    the neighbourhood structure of the document vectors genmodel.py then infers and indexes: top-10 neighbours by the index product."""
    from hiptagsearch.d2v import Doc2Vec
    from hiptagsearch.index import Similarity
    from test_oracle_d2v_train import _topic_corpus
    docs, labels = _topic_corpus(2400, topics=8, words_per_topic=40, doc_len=12, seed=5, shared=10)
    models = {}
    for mode in ("sequential", "parallel"):
        m = Doc2Vec(vector_size=64, window=50, min_count=1, workers=1, dm=0, sample=0, seed=3)
        m.build_vocab(docs)
        m.train(docs, total_examples=m.corpus_count, epochs=30, mode=mode)
        assert m.last_mode == mode and np.isfinite(m.syn1neg).all() and np.isfinite(m.doc_vectors).all()
        models[mode] = m
    assert models["sequential"].syn1neg.tobytes() != models["parallel"].syn1neg.tobytes()
    nbrs, purity = {}, {}
    for mode, m in models.items():
        vecs = m.inference().infer_vectors(docs, epochs=30)                                # genmodel.py:168-169
        idx = Similarity("t", None, 64, capacity=len(docs))
        idx.add_matrix(vecs)                                                                # genmodel.py:170-173: rows as inferred
        unit = vecs / np.linalg.norm(vecs, axis=1, keepdims=True)
        sims = idx.query(unit[:200])                                                        # webui.py:352 with unit queries
        top = np.argsort(-sims, axis=1)[:, :11]
        nbrs[mode] = [set(int(j) for j in row if j != i)for i, row in enumerate(top)]
        purity[mode] = float(np.mean([[labels[j] == labels[i] for j in list(nbrs[mode][i])[:10]] for i in range(200)]))
    overlap = float(np.mean([len(nbrs["sequential"][i] & nbrs["parallel"][i]) / 10.0 for i in range(200)]))
    print("topic purity of the top-10: sequential %.3f, parallel %.3f; neighbour overlap %.3f" % (purity["sequential"], purity["parallel"], overlap))
    assert purity["sequential"] > 0.9 and purity["parallel"] > 0.9
    assert purity["parallel"] > purity["sequential"] - 0.05
    assert overlap > 0.3                                       # 10 of 300 same-topic documents each: far above chance (10 / 2400 = 0.004)


def test_genmodel_cli_trains_the_model(tmp_path):
    """genmodel.py without --synthetic-d2v trains (genmodel.py:159-162), saves doc2vec_model and builds the index from inferred vectors."""
    from test_oracle_d2v_train import _topic_corpus
    docs, labels = _topic_corpus(120, topics=4, words_per_topic=12, doc_len=8, seed=2)
    with open(tmp_path / "tags-wd-tagger.txt", "w", encoding="utf-8") as f:
        for i, d in enumerate(docs):
            f.write("img%04d.png," % i + ",".join(d) + "\n")
    r = subprocess.run([sys.executable, os.path.join(PKG, "genmodel.py"), "--epochs", "30"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Doc2Vec trained: 120 documents, 53 tags, 30 epochs, sequential schedule" in r.stdout
    for f in ("doc2vec_model", "doc2vec_model.npz", "doc2vec_model.dv.npy", "doc2vec_index", "doc2vec_dictionary", "bm25_idf"):
        assert os.path.exists(tmp_path / f), f
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        from hiptagsearch import search
        eng = search.load_engine()
        # trained weights, not the zeros training starts from (with gensim's defaults -- sample = 1e-3, 300-d vectors starting at
        # +-1/300 -- a 120-document corpus moves them little in 30 epochs; the learning itself is checked in the tests above)
        assert np.abs(eng.model.syn1neg).max() > 1e-3
        res = eng.find_similar_documents("t1_w3 t1_w5", topn=20)
        assert len(res) >= 5 and np.mean([labels[d] == 1 for d, _ in res[:10]]) >= 0.8      # documents of topic 1 come first
    finally:
        os.chdir(cwd)
