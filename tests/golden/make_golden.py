#!/usr/bin/env python3
"""Generate golden input/output vectors by executing the REFERENCE's own
numpy / pure-Python arithmetic in the build container.

This script is test infrastructure.  It only runs where /root/reference is
mounted (the build container); the GPU box never sees the reference.  Only the
small data files it writes (inputs + expected outputs) are committed.

Technique: the reference's scripts import timm / gensim /
streamlit, which are not installed.  Their *own* arithmetic (BM25 build and
score, MCut, tag post-processing, result filter, batching loop) only needs
numpy, so
  * `tagging` and `genmodel` are imported with inert stub modules pre-seeded in
    sys.modules for the absent third-party packages;
  * for `webui.py` (runs Streamlit at import) only the wanted FunctionDefs are
    ast-extracted and exec'd into a namespace holding the module constants.

Run:  python -B tests/golden/make_golden.py
"""
import ast
import io
import json
import os
import sys
import tempfile
import types
import contextlib

import numpy as np

REF = os.environ.get("HIPTS_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# stubs for absent third-party modules
# --------------------------------------------------------------------------
def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Dictionary:
    """Minimal stand-in for gensim.corpora.Dictionary: token2id in first-seen
    order per document with tokens of one document sorted (gensim assigns ids
    to the *sorted* new tokens of each document) [3P-mem]."""

    def __init__(self, docs=None):
        self.token2id = {}
        for d in docs or []:
            for tok in sorted(set(d)):
                if tok not in self.token2id:
                    self.token2id[tok] = len(self.token2id)


def install_stubs():
    if "timm" not in sys.modules:
        _stub("timm")
        _stub("timm.data", create_transform=lambda **k: None, resolve_data_config=lambda *a, **k: {})
        sys.modules["timm"].data = sys.modules["timm.data"]
    if "gensim" not in sys.modules:
        g = _stub("gensim")
        corpora = _stub("gensim.corpora", Dictionary=_Dictionary)
        g.corpora = corpora
        models = _stub("gensim.models", Doc2Vec=object)
        g.models = models
        d2v = _stub("gensim.models.doc2vec", TaggedDocument=lambda words, tags: (words, tags))
        models.doc2vec = d2v
        sims = _stub("gensim.similarities", Similarity=object, MatrixSimilarity=object)
        g.similarities = sims


def load_webui_functions(names):
    """ast-extract FunctionDefs `names` from webui.py and exec them in a
    namespace holding the constants they use."""
    src = open(os.path.join(REF, "webui.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    consts = [n for n in tree.body
              if isinstance(n, (ast.Assign, ast.AnnAssign))
              and any(isinstance(t, ast.Name) and t.id in (
                  "BM25_WEIGHT", "DOC2VEC_WEIGHT", "ORIGINAL_SCORE_WEIGHT", "RERANKED_SCORE_WEIGHT",
                  "DIFF_FILTER_THRESH", "REQUIRE_TAG_MAGIC_NUMBER", "NG_WORDS")
                      for t in ([n.target] if isinstance(n, ast.AnnAssign) else n.targets))]
    mod = ast.Module(body=consts + wanted, type_ignores=[])
    ns = {"np": np, "math": __import__("math")}
    from typing import List, Tuple, Dict, Any, Optional
    ns.update(List=List, Tuple=Tuple, Dict=Dict, Any=Any, Optional=Optional, ndarray=np.ndarray)
    exec(compile(mod, "webui_extract", "exec"), ns)
    return ns


# --------------------------------------------------------------------------
# synthetic corpora (same generator the tests / bench use: hiptagsearch.synth)
# --------------------------------------------------------------------------
def synth_docs(D, V, seed, mean_len=20, repeat_frac=0.01):
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, V + 1, dtype=np.float64)
    p = ranks ** -1.1
    p /= p.sum()
    docs = []
    for d in range(D):
        n = int(np.clip(rng.poisson(mean_len), 3, 60))
        n = min(n, V)
        ids = rng.choice(V, size=n, replace=False, p=p)
        toks = ["t%05d" % i for i in ids]
        if rng.random() < repeat_frac:
            k = int(rng.integers(1, 3))
            toks += [toks[0]] * k
        docs.append(toks)
    return docs


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    install_stubs()
    cwd0 = os.getcwd()
    work = tempfile.mkdtemp(prefix="hipts_golden_")
    os.chdir(work)
    try:
        import genmodel  # reference module (numpy arithmetic only is used)
        import tagging   # reference module
        import pickle

        # ---------------- G1: BM25 build -------------------------------------
        g1 = {}
        for name, docs in (
            ("tiny", [["a", "b", "c"], ["a", "d", "d", "e"], ["b", "d", "f", "g", "h"], ["x", "y", "z"]]),
            ("d1000", synth_docs(1000, 300, seed=7)),
        ):
            dic = _Dictionary(docs)
            with contextlib.redirect_stdout(io.StringIO()):
                genmodel.gen_and_save_bm25_index([list(d) for d in docs], dic)
            corpus = pickle.load(open("bm25_corpus", "rb"))
            idf = pickle.load(open("bm25_idf", "rb"))
            avgdl = pickle.load(open("bm25_avgdl", "rb"))
            Dn = pickle.load(open("bm25_D", "rb"))
            dl = pickle.load(open("bm25_doc_lengths", "rb"))
            g1[name] = {
                "docs": docs,
                "token2id": dic.token2id,
                "corpus": [{str(k): int(v) for k, v in d.items()} for d in corpus],
                "idf_keys": [int(k) for k in idf.keys()],
                "idf_hex": [float(v).hex() for v in idf.values()],
                "idf_type": type(next(iter(idf.values()))).__name__,
                "avgdl_hex": float(avgdl).hex(),
                "avgdl_type": type(avgdl).__name__,
                "D": int(Dn),
                "D_type": type(Dn).__name__,
                "doc_lengths": [int(x) for x in dl],
                "doc_lengths_dtype": str(dl.dtype),
            }
        json.dump(g1, open(os.path.join(OUT, "g1_bm25_build.json"), "w"))

        # ---------------- G2: BM25 score --------------------------------------
        ns = load_webui_functions({"compute_bm25_scores", "filter_searched_result"})
        g2 = {}
        for name in ("tiny", "d1000"):
            docs = g1[name]["docs"]
            dic = _Dictionary(docs)
            with contextlib.redirect_stdout(io.StringIO()):
                genmodel.gen_and_save_bm25_index([list(d) for d in docs], dic)
            ns["bm25_corpus"] = pickle.load(open("bm25_corpus", "rb"))
            ns["bm25_idf"] = pickle.load(open("bm25_idf", "rb"))
            ns["bm25_avgdl"] = pickle.load(open("bm25_avgdl", "rb"))
            ns["bm25_D"] = pickle.load(open("bm25_D", "rb"))
            ns["bm25_doc_lengths"] = pickle.load(open("bm25_doc_lengths", "rb"))
            ns["dictionary"] = dic
            t2i = dic.token2id
            rng = np.random.default_rng(11)
            queries = []
            if name == "tiny":
                queries = [
                    {t2i["a"]: 1, t2i["d"]: 1002, t2i["x"]: -1},
                    {t2i["a"]: 1},
                    {t2i["d"]: 3},
                    {t2i["b"]: 1001},
                    {t2i["z"]: -2, t2i["b"]: 2},
                    {9999: 1},                      # id unknown to idf -> idf 0
                    {9999: 1001},                   # required but absent everywhere -> all -inf
                    {t2i["a"]: 1000},               # weight == 1000 is NOT "required" (strict >)
                ]
            else:
                V = len(t2i)
                for _ in range(40):
                    nt = int(rng.integers(1, 5))
                    ids = rng.choice(min(V, 60), size=nt, replace=False)
                    q = {}
                    for t in ids:
                        r = rng.random()
                        w = int(rng.integers(1, 4))
                        q[int(t)] = w if r < 0.7 else (1000 + w if r < 0.85 else -w)
                    queries.append(q)
            outs = []
            for q in queries:
                s = ns["compute_bm25_scores"](query_weights=dict(q))
                assert s.dtype == np.float64
                outs.append([float(x).hex() for x in s])
            g2[name] = {"queries": [[[int(k), int(v)] for k, v in q.items()] for q in queries], "scores_hex": outs}
        json.dump(g2, open(os.path.join(OUT, "g2_bm25_score.json"), "w"))

        # ---------------- G3: mcut_threshold ----------------------------------
        rng = np.random.default_rng(3)
        g3 = []
        cases = [np.array([.9, .85, .3, .28, .05]),
                 np.array([.5, .5, .5, .1]),
                 np.array([.7, .4, .1]),               # equal gaps -> first
                 np.array([.2, .9]),
                 np.array([1.0, 1.0, 0.0, 0.0])]
        for n in (7, 64, 1000, 8257):
            x = rng.random(n).astype(np.float32).astype(np.float64)
            cases.append(x)
            lg = (rng.standard_normal(n) * 3).astype(np.float32)
            cases.append((1 / (1 + np.exp(-lg.astype(np.float32)))).astype(np.float32).astype(np.float64))
        for c in cases:
            t = tagging.mcut_threshold(np.array(c, dtype=np.float64))
            g3.append({"probs_hex": [float(v).hex() for v in c], "thresh_hex": float(t).hex()})
        json.dump(g3, open(os.path.join(OUT, "g3_mcut.json"), "w"))

        # ---------------- G4: Predictor.predict post-processing ---------------
        import torch
        C = 600
        cat = np.zeros(C, dtype=np.int64)
        cat[:4] = 9
        cat[4:450] = 0
        cat[450:] = 4
        names = ["tag %04d" % i for i in range(C)]
        names[10] = "^_^"
        names[11] = "long hair"
        rng = np.random.default_rng(4)
        logits = (rng.standard_normal((8, C)) * 2.5).astype(np.float32)
        logits[1, 4:450] = -8.0
        logits[1, 20] = 5.0
        logits[1, 21] = 4.0       # two confident general tags
        logits[2, 450:] = -9.0    # no character above 0.15
        logits[3, 30] = logits[3, 31] = logits[3, 32] = 6.0   # exact prob ties -> label order kept
        logits[4, :] = 0.0        # all probs equal -> gaps all 0 -> argmax 0 -> nothing selected

        class FakeModel:
            def __init__(self, out):
                self.out = out

            def forward(self, x):
                return self.out[: x.shape[0]]

        P = tagging.Predictor()
        P.tagger_model = FakeModel(torch.from_numpy(logits))
        P.tag_names = names
        P.rating_index = list(np.where(cat == 9)[0])
        P.general_index = list(np.where(cat == 0)[0])
        P.character_index = list(np.where(cat == 4)[0])
        tensors = [torch.zeros(3, 4, 4) for _ in range(8)]
        with contextlib.redirect_stdout(io.StringIO()):
            lines = P.predict(tensors, 0.3, True, 0.3, True)
        np.savez_compressed(os.path.join(OUT, "g4_predict.npz"), logits=logits, category=cat)
        json.dump({"names": names, "lines": lines}, open(os.path.join(OUT, "g4_predict.json"), "w"))

        # ---------------- G5: filter_searched_result --------------------------
        f = ns["filter_searched_result"]
        cases5 = [
            [(3, .9), (1, .9), (0, .5), (2, .5 - 1e-9), (4, .1), (5, 0.0)],
            [(0, 1.0), (1, .8), (2, .6), (3, .2)],
            [(0, 1.0), (1, .8), (2, .8 - 1e-8), (3, .2)],
            [(0, 1.0), (1, .8), (2, .8 - 1e-8), (3, .2), (4, .2 - 1e-7), (5, .1)],
            [(7, 2.0), (8, 1.0), (9, 0.0), (10, -1.0)],
            [(0, 1.0), (1, 1.0), (2, 1.0)],
            [(0, .5), (1, .25)],
        ]
        g5 = []
        for c in cases5:
            r = f(list(c))
            g5.append({"in": [[int(a), float(b).hex()] for a, b in c],
                       "out": [[int(a), float(b).hex()] for a, b in r]})
        json.dump(g5, open(os.path.join(OUT, "g5_filter.json"), "w"))

        # ---------------- G6: process_directory batching loop -----------------
        g6 = []
        for N in (5, 10, 11, 25, 30, 32, 41):
            d = tempfile.mkdtemp(prefix="imgs_", dir=work)
            for i in range(N):
                open(os.path.join(d, "im%03d.png" % i), "wb").close()
            P = tagging.Predictor()
            P.load_model = lambda: None
            P.gen_image_tensor = lambda path: path          # "tensor" = the path itself
            P.predict = lambda tensors, a, b, c, e: ["tagA,tagB" for _ in tensors]
            if os.path.exists("tags-wd-tagger.txt"):
                os.remove("tags-wd-tagger.txt")
            with contextlib.redirect_stdout(io.StringIO()):
                P.process_directory(d)
            P.f.close()
            lines = open("tags-wd-tagger.txt", encoding="utf-8").read().splitlines()
            listed = P.list_files_recursive(d)
            order = [os.path.basename(x) for x in listed]
            got = [os.path.basename(l.split(",")[0]) for l in lines]
            # batches: consecutive groups of 10 lines hold exactly the files of one submitted batch
            batches_ok = all(sorted(got[i:i + 10]) == sorted(order[i:i + 10]) for i in range(0, len(got), 10))
            g6.append({"N": N, "lines": len(lines), "batch_sets_match_listing_order": bool(batches_ok)})
        json.dump(g6, open(os.path.join(OUT, "g6_batching.json"), "w"))

        # ---------------- G8: read_documents_and_gen_idx_text -----------------
        txt = ("p0.png,a,b,c\n" "p1.png,a,b\n" "p2.png\n" "p3.png,x,y,z,w\n" "\n" "p,4.png,q,r\n" " p5.png,k,l,m \n")
        open("tags-wd-tagger.txt", "w", encoding="utf-8").write(txt)
        docs, tagged = genmodel.read_documents_and_gen_idx_text("tags-wd-tagger.txt")
        idx = open("tags-wd-tagger_doc2vec_idx.csv", encoding="utf-8").read()
        json.dump({"input": txt, "docs": docs, "idx_text": idx}, open(os.path.join(OUT, "g8_read_documents.json"), "w"))

        # ---------------- G7: prepare_image -----------------------------------
        from PIL import Image
        rng = np.random.default_rng(9)
        g7 = {}
        P = tagging.Predictor()
        for mode, shape in (("RGBA", (5, 9)), ("LA", (8, 3)), ("RGB", (6, 6)), ("L", (4, 7)), ("P", (7, 2))):
            nch = {"RGBA": 4, "LA": 2, "RGB": 3, "L": 1, "P": 1}[mode]
            arr = rng.integers(0, 256, (shape[0], shape[1], nch), dtype=np.uint8)
            img = Image.fromarray(arr.squeeze() if nch == 1 else arr, mode if mode != "P" else "L")
            if mode == "P":
                img = img.convert("P")
            out = np.asarray(P.prepare_image(img))
            g7["in_" + mode] = np.asarray(img.convert("RGBA") if mode == "P" else img)
            g7["mode_" + mode] = np.array(mode)
            g7["out_" + mode] = out
        np.savez_compressed(os.path.join(OUT, "g7_prepare_image.npz"), **g7)
        print("golden fixtures written to", OUT)
    finally:
        os.chdir(cwd0)


if __name__ == "__main__":
    main()
