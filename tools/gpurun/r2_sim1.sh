#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_configs.py tests/test_gpu_flows.py -m gpu -q -x 2>&1 | tail -3 || exit 1
timeout -k 10 200 python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "anime-illust-image-searcher_amd")
from hiptagsearch.index import Similarity
rng = np.random.default_rng(45)
for K in (300, 768):
    rows = rng.standard_normal((100_000, K)).astype(np.float32)
    idx = Similarity("b", None, K, capacity=100_000); idx.add_matrix(rows)
    q = rows[:64].copy()
    import torch
    out = torch.empty((1, 100_000), dtype=torch.float32, device="cuda")
    for i in range(4): idx.query(q[i])
    t0 = time.perf_counter()
    for i in range(64): idx.query(q[i])
    a = (time.perf_counter() - t0) / 64 * 1e6
    t0 = time.perf_counter()
    for i in range(64): idx.query(q[i], out=out)
    torch.cuda.synchronize()
    b = (time.perf_counter() - t0) / 64 * 1e6
    print("dim %d: one query over 100k rows: %.1f us with the scores copied to the host, %.1f us device-resident" % (K, a, b))
PY
