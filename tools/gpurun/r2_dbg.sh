#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 120 python - > gpurun_out/dbg.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "anime-illust-image-searcher_amd"); sys.path.insert(0, ".")
import numpy as np
from hiptagsearch import synth
from hiptagsearch.d2v import Doc2VecInference
from oracle import d2v as od2v
dim, V, ndocs, epochs = 64, 500, 8, 3
ptr, terms = synth.tag_corpus(D=ndocs, V=V, seed=7)
m = synth.d2v_model(synth.term_counts(ptr, terms, V), dim=dim, seed=44)
v0, seeds = synth.d2v_inputs(ndocs, dim, seed=44)
model = Doc2VecInference(m["syn1neg"], m["cum_table"], m["sample_int"], {}, epochs=epochs)
print("launching", flush=True)
got = model.infer_batch(ptr, terms, v0, seeds)
want = od2v.infer(m["syn1neg"], m["cum_table"], m["sample_int"], ptr, terms, v0, seeds, epochs)
print("equal:", got.tobytes() == want.tobytes(), np.abs(got - want).max())
PY
rc=$?; echo rc=$rc; grep -vE "^\s+File|Extension modules" gpurun_out/dbg.log | head -20 | cut -c1-400
grep -q "Memory access fault" gpurun_out/dbg.log && exit 1; [ $rc -eq 0 ] || exit 1; ./tools/gpurun/r2_d2v.sh
