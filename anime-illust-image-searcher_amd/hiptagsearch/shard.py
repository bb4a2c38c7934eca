"""Multi-GPU sharding of the tagging stage: one process per GPU, images are
independent units.  Rank r tags a contiguous block of the file list; each rank produces fixed-width
int32 tag rows {n_general, n_character, ids...}; ONE all-gather (RCCL over xGMI on GPUs, gloo in the
CPU tests) concatenates them in rank order, which is file order.  No other collective on the path.
"""
from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: every rank gets ceil(n/world) items except the tail."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def padded_rows_per_rank(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def gather_rows(local_rows, n_items: int, dist=None):
    """All-gather equal-sized row blocks in rank order and drop the sentinel padding.
    local_rows: torch int32 tensor [padded_rows_per_rank, width] (rows beyond the rank's share are
    sentinels, n_general = -1).  Returns a tensor [n_items, width] on every rank."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rows[:n_items]
    world = dist.get_world_size()
    out = torch.empty((world * local_rows.shape[0], local_rows.shape[1]), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, local_rows.contiguous())
    per = local_rows.shape[0]
    keep = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        keep.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(keep, dim=0)


def rows_to_lines(rows: np.ndarray, names: Sequence[str], paths: Sequence[str]) -> List[str]:
    """tags-wd-tagger.txt lines (tagging.py:335: path + ',' + tags) from gathered rows."""
    lines = []
    width = rows.shape[1] - 2
    for p, row in zip(paths, rows):
        ng, nc = int(row[0]), int(row[1])
        n = min(ng + nc, width)
        lines.append(p + "," + ",".join(names[i].replace(" ", "_") for i in row[2:2 + n]))
    return lines
