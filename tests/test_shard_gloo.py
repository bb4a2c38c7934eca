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
