"""smoke(): one small invocation of every stage of the hot path on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    import hiptagsearch  # noqa: F401
    from hiptagsearch import synth
    from hiptagsearch.bm25 import BM25Index
    from hiptagsearch.index import Similarity
    from hiptagsearch.search import SearchEngine
    from hiptagsearch.tagger import TagSelector, ViTTagger
    from oracle import bm25 as obm25, search as osearch, tags as otags, vit as ovit

    # tagging: tiny ViT forward + tag selection
    cfg = dict(synth.VIT_TINY)
    w = synth.vit_weights(cfg, seed=1)
    imgs = synth.images_u8(3, cfg["image_size"], seed=2)
    model = ViTTagger(cfg, w, max_batch=4)
    logits, probs = model.forward_u8(imgs)
    want = ovit.vit_forward(ovit.to_torch(w), ovit.preprocess_u8_nhwc(imgs), patch=cfg["patch"], heads=cfg["heads"]).numpy()
    err = float(np.abs(logits - want).max())
    assert err <= 1e-3, "ViT logits differ from the oracle by %g" % err
    names, cat = synth.label_table(cfg["num_classes"])
    counts, ids, _ = TagSelector(cat, 4).run(probs)
    gi, ci = list(np.where(cat == 0)[0]), list(np.where(cat == 4)[0])
    for r in range(3):
        g, c, _, _ = otags.select_indices(probs[r], gi, ci)
        assert ids[r, :len(g) + len(c)].tolist() == g + c
    # query: BM25 + index product + combine + top-k
    V, D = 500, 3000
    ptr, terms = synth.tag_corpus(D, V, seed=1)
    rows = synth.index_vectors(D, 300, seed=2)
    bm = BM25Index(ptr, terms, V)
    idx = Similarity("smoke", None, 300, capacity=D)
    idx.add_matrix(rows)
    q = {3: 1.0, 7: 1002.0, 11: -1.0}
    qv = rows[5] / np.linalg.norm(rows[5])
    gids, gvals = SearchEngine(None, idx, {}, bm, []).score_topk([q], qv[None], 50)
    e = bm.export()
    b = obm25.bm25_score_csr(e["csr_ptr"], e["csr_term"], e["csr_tf"], e["idf"], bm.avgdl, e["doc_len"], list(q.keys()), list(q.values()))
    f = osearch.combine(b, osearch.similarity(rows, qv.astype(np.float32)))
    wi, wv = osearch.topk(f, 50)
    assert np.array_equal(gids[0], wi) and gvals[0].tobytes() == wv.tobytes(), "top-k differs from the oracle"
    # CCIP feature encoder (CAFormer): tiny geometry vs the oracle
    from hiptagsearch.cfeatures import CCIPEncoder
    from oracle import ccip as occip
    ccfg = dict(synth.CCIP_TINY)
    cw = synth.ccip_weights(ccfg, seed=3)
    cimgs = synth.images_u8(2, ccfg["image_size"], seed=47)
    feat = CCIPEncoder(ccfg, cw, max_batch=2).forward_u8(cimgs)
    cwant = occip.metaformer_forward(occip.to_torch(cw), occip.preprocess_u8_nhwc(cimgs), dims=ccfg["dims"], depths=ccfg["depths"]).numpy()
    cerr = float(np.abs(feat - cwant).max())
    assert cerr <= 1e-1, "CCIP features differ from the oracle by %g" % cerr
    # EVA02 tagger (the reference's real model family): tiny geometry vs the oracle
    from hiptagsearch.tagger import EvaTagger
    from oracle import eva as oeva
    ecfg = dict(synth.EVA02_TINY)
    ew = synth.eva_weights(ecfg, seed=1)
    eimgs = synth.images_u8(2, ecfg["image_size"], seed=2)
    elog, _ = EvaTagger(ecfg, ew, max_batch=2).forward_u8(eimgs)
    ewant = oeva.eva_forward(oeva.to_torch(ew), ovit.preprocess_u8_nhwc(eimgs), patch=ecfg["patch"], heads=ecfg["heads"]).numpy()
    eerr = float(np.abs(elog - ewant).max())
    assert eerr <= 1e-3, "EVA02 logits differ from the oracle by %g" % eerr       # IEEE-half operands (EvaTagger default)
    print("smoke ok: ViT max|dlogit| = %.2e, tag rows and top-50 identical to the oracle, CCIP max|df| = %.2e, EVA02 max|dlogit| = %.2e" % (err, cerr, eerr))
