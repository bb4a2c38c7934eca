#!/usr/bin/env python3
"""Development aid: Doc2Vec inference latency (1 and 10 documents: the query function's per-tag inference and rerank) and throughput."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np, torch
from hiptagsearch import synth
from hiptagsearch.d2v import Doc2VecInference
D, V, K = 100_000, 10_000, 300
ptr, terms = synth.tag_corpus(D, V, seed=42)
m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=K, seed=44)
v0, seeds = synth.d2v_inputs(D, K, seed=44)
model = Doc2VecInference(m["syn1neg"], m["cum_table"], m["sample_int"], {}, epochs=100)
def run(n):
    return model.infer_batch(ptr[:n + 1], terms[:ptr[n]], v0[:n], seeds[:n])
run(64)
for n in (1, 10):
    t0 = time.perf_counter()
    for _ in range(5): run(n)
    print("plan=%s  %2d documents x 100 epochs: %.2f ms per call" % (os.environ.get("HIPTS_D2V_PLAN", "1"), n, (time.perf_counter() - t0) / 5 * 1e3))
one = np.array([0, 1], dtype=np.int64)
t0 = time.perf_counter()
for _ in range(5): model.infer_batch(one, terms[:1], v0[:1], seeds[:1])
print("plan=%s  one-word document (a query tag, webui.py:106): %.2f ms" % (os.environ.get("HIPTS_D2V_PLAN", "1"), (time.perf_counter() - t0) / 5 * 1e3))
t0 = time.perf_counter(); run(20000); dt = time.perf_counter() - t0
print("plan=%s  20000 documents: %.0f documents/s" % (os.environ.get("HIPTS_D2V_PLAN", "1"), 20000 / dt))
