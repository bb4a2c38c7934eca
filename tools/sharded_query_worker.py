#!/usr/bin/env python3
"""Test helper: one rank of the sharded query path (tests/test_gpu_query.py launches WORLD_SIZE of these on
the one GPU of the box with the gloo backend; on a multi-GPU node the same code runs over RCCL with
--backend nccl and one GPU per rank).  Rank 0 checks the merged top-k against the unsharded engine."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--docs", type=int, default=5000)
    ap.add_argument("--k", type=int, default=100)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    device = int(os.environ.get("LOCAL_RANK", 0)) if args.backend == "nccl" else 0
    torch.cuda.set_device(device)
    from hiptagsearch import synth
    from hiptagsearch.shard import ShardedSearchEngine
    V, D, K = 800, args.docs, 300
    ptr, terms = synth.tag_corpus(D, V, seed=7)
    rows = synth.index_vectors(D, K, seed=8)
    qs = [dict(q) for q in synth.queries(24, V, seed=9)]
    rng = np.random.default_rng(10)
    qv = rng.standard_normal((len(qs), K)).astype(np.float32)
    eng = ShardedSearchEngine(ptr, terms, V, rows, rank, world, device=device)
    ids, vals = eng.score_topk(qs, qv, args.k, dist=dist)
    ok = 1
    if rank == 0:
        from hiptagsearch.bm25 import BM25Index
        from hiptagsearch.index import Similarity
        from hiptagsearch.search import SearchEngine
        bm = BM25Index(ptr, terms, V, device)
        idx = Similarity("whole", None, K, device, capacity=D)
        idx.add_matrix(rows)
        wi, wv = SearchEngine(None, idx, {}, bm, []).score_topk(qs, qv, args.k)
        ok = int(np.array_equal(ids, wi.astype(np.int64)) and vals.tobytes() == wv.tobytes())
        print("sharded query world=%d: %s" % (world, "identical to the unsharded engine" if ok else "MISMATCH"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
