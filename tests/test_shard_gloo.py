"""CPU, world_size 2 over gloo: the N>1 path of the tagging stage (contiguous sharding + one
all-gather of fixed-width tag rows restoring file order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, width, q):
    sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the shard module is pure host logic; import it without loading the HIP library
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard", os.path.join(ROOT, "anime-illust-image-searcher_amd", "hiptagsearch", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    lo, hi = shard.shard_range(n_items, rank, world)
    per = shard.padded_rows_per_rank(n_items, world)
    rows = torch.full((per, width), -1, dtype=torch.int32)
    for i in range(lo, hi):                     # a rank's "tagging": row i = {i % 7, i % 3, i, i+1, ...}
        rows[i - lo, 0] = i % 7
        rows[i - lo, 1] = i % 3
        rows[i - lo, 2:] = torch.arange(i, i + width - 2, dtype=torch.int32)
    full = shard.gather_rows(rows, n_items, dist)
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [10, 7, 1, 64])
def test_two_rank_gather_restores_file_order(n_items):
    world, width = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.concatenate([[i % 7, i % 3], np.arange(i, i + width - 2)]) for i in range(n_items)]).astype(np.int32)
    for r in range(world):
        np.testing.assert_array_equal(results[r], want)


def test_shard_ranges_cover_everything():
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard", os.path.join(ROOT, "anime-illust-image-searcher_amd", "hiptagsearch", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    for n in (0, 1, 7, 8, 9, 100_000, 1_000_003):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) <= shard.padded_rows_per_rank(n, world)
    names = ["a b", "c", "d"]
    rows = np.array([[1, 1, 0, 2, 9], [0, 0, 9, 9, 9]], dtype=np.int32)
    assert shard.rows_to_lines(rows, names, ["p0", "p1"]) == ["p0,a_b,d", "p1,"]


# ---------------------------------------------------------------------------------------------
# Sharded query path: host pieces (global statistics, k-way merge) and the collective plumbing
# ---------------------------------------------------------------------------------------------
def _load_shard():
    import importlib.util
    spec = importlib.util.spec_from_file_location("shard", os.path.join(ROOT, "anime-illust-image-searcher_amd", "hiptagsearch", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    return shard


def test_global_bm25_stats_match_the_oracle():
    """The statistics every shard scores with are those of the whole corpus, bit for bit (genmodel.py:69-82)."""
    sys.path.insert(0, ROOT)
    from oracle import bm25 as obm25
    shard = _load_shard()
    rng = np.random.default_rng(0)
    V, D = 50, 300
    docs = [[("t%d" % t) for t in rng.integers(0, V + 5, rng.integers(1, 12))] for _ in range(D)]     # ids >= V are out of vocabulary
    token2id = {"t%d" % t: t for t in range(V)}
    corpus, idf, avgdl, D_, dl = obm25.bm25_build(docs, token2id)
    ptr = np.cumsum([0] + [len(d) for d in docs]).astype(np.int64)
    ids = np.array([token2id.get(t, -1) for d in docs for t in d], dtype=np.int32)
    got_idf, got_avgdl = shard.global_bm25_stats(ptr, ids, V)
    assert float(got_avgdl).hex() == float(avgdl).hex()
    for t, v in idf.items():
        assert float(got_idf[t]).hex() == float(v).hex()
    assert all(got_idf[t] == 0.0 for t in range(V) if t not in idf)


def test_merge_topk_is_the_global_order():
    shard = _load_shard()
    rng = np.random.default_rng(1)
    nq, D, k, world = 5, 400, 37, 3
    vals = np.round(rng.standard_normal((nq, D)), 1)          # many exact ties
    vals[:, ::17] = -np.inf
    cv, ci = [], []
    for r in range(world):
        lo, hi = shard.shard_range(D, r, world)
        order = np.stack([np.lexsort((np.arange(lo, hi), -vals[q, lo:hi]))[:k] for q in range(nq)])
        ci.append(order + lo)
        cv.append(np.take_along_axis(vals[:, lo:hi], order, axis=1))
    ids, got = shard.merge_topk(cv, ci, k)
    for q in range(nq):
        want = np.lexsort((np.arange(D), -vals[q]))[:k]
        assert ids[q].tolist() == want.tolist()
        assert got[q].tobytes() == vals[q][want].tobytes()


def _query_worker(rank, world, port, q):
    """score_topk's collective sequence with the device phases replaced by numpy: all-reduce(MAX) of the
    row maxima, all-gather of padded candidates, merge."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = _load_shard()
    rng = np.random.default_rng(3)
    nq, D, k = 4, 101, 20
    a = np.abs(rng.standard_normal((nq, D)))
    b = rng.standard_normal((nq, D)).astype(np.float32)

    class Fake(shard.ShardedSearchEngine):
        def __init__(self):
            self.rank, self.world, self.D = rank, world, D
            self.lo, self.hi = shard.shard_range(D, rank, world)

        def local_scores(self, qw, qv):
            la, lb = torch.from_numpy(a[:, self.lo:self.hi].copy()), torch.from_numpy(b[:, self.lo:self.hi].copy())
            return la, lb, la.max(dim=1).values, lb.max(dim=1).values

        def local_topk(self, bm, sims, max_a, max_b, k, w_bm25=0.5, w_sim=0.5):
            final = 0.5 * (bm / max_a[:, None]).numpy() + (np.float32(0.5) * (sims / max_b[:, None]).numpy()).astype(np.float64)
            kk = min(k, final.shape[1])
            order = np.stack([np.lexsort((np.arange(final.shape[1]), -final[i]))[:kk] for i in range(final.shape[0])])
            return np.take_along_axis(final, order, axis=1), order.astype(np.int64) + self.lo

    ids, vals = Fake().score_topk([{}] * nq, np.zeros((nq, 1), np.float32), k, dist=dist)
    final = 0.5 * (a / a.max(axis=1, keepdims=True)) + (np.float32(0.5) * (b / b.max(axis=1, keepdims=True))).astype(np.float64)
    want = np.stack([np.lexsort((np.arange(D), -final[i]))[:k] for i in range(nq)])
    q.put((rank, bool(np.array_equal(ids, want) and np.array_equal(vals, np.take_along_axis(final, want, axis=1)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_query_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_query_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}
