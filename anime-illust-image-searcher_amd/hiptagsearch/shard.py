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
    if dist.get_backend() != "nccl":            # gloo (tests, ranks sharing a GPU) moves host tensors; RCCL moves device tensors
        local_rows = local_rows.cpu()
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


# ---------------------------------------------------------------------------------------------
# Query path over an index whose documents are sharded across ranks (SURVEY.md section 8e,
# "Partitioning (query)"): rank r holds the BM25 postings and the dense index rows of a contiguous
# block of documents, scored with the statistics of the WHOLE corpus.  Per query batch:
#   local BM25 + index product  ->  local row maxima  ->  all-reduce(MAX) of 2 numbers per query
#   ->  normalise + combine with the GLOBAL maxima  ->  local top-k  ->  all-gather of k
#   (score, global id) pairs per rank  ->  k-way merge (score descending, ties by ascending id).
# Every document's combined score is computed by the same instructions on the same inputs as in the
# unsharded engine, so the merged top-k is bit-identical to SearchEngine.score_topk.
# ---------------------------------------------------------------------------------------------
def global_bm25_stats(doc_ptr: np.ndarray, term_ids: np.ndarray, vocab: int):
    """(idf float64 [vocab], avgdl) of the whole corpus with the reference's expressions
    (genmodel.py:69-82): df counts documents containing a term, dl counts in-vocabulary tokens."""
    doc_ptr = np.asarray(doc_ptr, dtype=np.int64)
    term_ids = np.asarray(term_ids, dtype=np.int32)
    D = len(doc_ptr) - 1
    valid = (term_ids >= 0) & (term_ids < vocab)
    doc_of = np.repeat(np.arange(D, dtype=np.int64), np.diff(doc_ptr))
    dl = np.bincount(doc_of[valid], minlength=D).astype(np.int64)
    pairs = np.unique(doc_of[valid] * np.int64(vocab) + term_ids[valid].astype(np.int64))
    df = np.bincount((pairs % vocab).astype(np.int64), minlength=vocab)
    idf = np.zeros(vocab, dtype=np.float64)
    for t in np.nonzero(df)[0]:
        idf[t] = np.log(1 + (D - int(df[t]) + 0.5) / (int(df[t]) + 0.5))              # genmodel.py:81
    return idf, np.float64(np.mean(dl)) if D else np.float64("nan")                     # genmodel.py:76


def merge_topk(vals_list: Sequence[np.ndarray], ids_list: Sequence[np.ndarray], k: int) -> Tuple[np.ndarray, np.ndarray]:
    """k-way merge of per-rank candidates [nq, k_r] (score float64, GLOBAL id) into [nq, k]: score
    descending, ties by ascending id -- the order of the unsharded top-k (webui.py:191-192 on the full list)."""
    vals = np.concatenate(vals_list, axis=1)
    ids = np.concatenate(ids_list, axis=1).astype(np.int64)
    nq = vals.shape[0]
    kk = min(k, vals.shape[1])
    out_v = np.empty((nq, kk), dtype=np.float64)
    out_i = np.empty((nq, kk), dtype=np.int64)
    for q in range(nq):
        order = np.lexsort((ids[q], -vals[q]))[:kk]
        out_v[q], out_i[q] = vals[q][order], ids[q][order]
    return out_i, out_v


class ShardedSearchEngine:
    """One rank's shard of the query path.  `doc_ptr/term_ids` describe the WHOLE corpus (they are only
    used on the host to take the global statistics and to cut this rank's block), `rows` the whole
    dense matrix or just this rank's block (`rows_are_local`)."""

    def __init__(self, doc_ptr: np.ndarray, term_ids: np.ndarray, vocab: int, rows: np.ndarray, rank: int, world: int,
                 device: int = 0, rows_are_local: bool = False, stats=None):
        from .bm25 import BM25Index
        from .index import Similarity
        from . import _lib
        self._lib = _lib
        doc_ptr = np.asarray(doc_ptr, dtype=np.int64)
        self.D = len(doc_ptr) - 1
        self.rank, self.world, self.device = rank, world, device
        self.lo, self.hi = shard_range(self.D, rank, world)
        idf, avgdl = stats if stats is not None else global_bm25_stats(doc_ptr, term_ids, vocab)
        lp = doc_ptr[self.lo:self.hi + 1] - doc_ptr[self.lo]
        lt = np.asarray(term_ids, dtype=np.int32)[doc_ptr[self.lo]:doc_ptr[self.hi]]
        self.bm25 = BM25Index(lp, lt, vocab, device, numpy_idf=False)
        self.bm25.set_idf(idf)                                              # global statistics, not the shard's
        _lib.call("hipts_bm25_set_avgdl", self.bm25._h, ctypes_double(avgdl))
        local_rows = rows if rows_are_local else rows[self.lo:self.hi]
        self.index = Similarity("shard%d" % rank, None, int(local_rows.shape[1]), device, capacity=max(1, self.hi - self.lo))
        self.index.add_matrix(np.ascontiguousarray(local_rows, dtype=np.float32))

    # -- phase 1: local scores and local maxima (device tensors)
    def local_scores(self, query_weights: Sequence[dict], query_vectors: np.ndarray):
        import torch
        nq, Dl = len(query_weights), self.hi - self.lo
        dev = "cuda:%d" % self.device
        bm = torch.empty((nq, Dl), dtype=torch.float64, device=dev)
        self.bm25.score(query_weights, out=bm)
        sims = torch.empty((nq, Dl), dtype=torch.float32, device=dev)
        self.index.query(np.ascontiguousarray(np.atleast_2d(query_vectors), dtype=np.float32), out=sims)
        max_a = torch.empty(nq, dtype=torch.float64, device=dev)
        max_b = torch.empty(nq, dtype=torch.float32, device=dev)
        L = self._lib
        L.call("hipts_rowmax", L.ptr(bm), L.ptr(sims), nq, ctypes_int64(Dl), L.ptr(max_a), L.ptr(max_b), self.device, L.current_stream_ptr())
        return bm, sims, max_a, max_b

    # -- phase 2: combine with the global maxima, local top-k with GLOBAL ids
    def local_topk(self, bm, sims, max_a, max_b, k: int, w_bm25: float = 0.5, w_sim: float = 0.5):
        import torch
        L = self._lib
        nq, Dl = bm.shape
        final = torch.empty_like(bm)
        L.call("hipts_combine_with_max", L.ptr(bm), L.ptr(sims), nq, ctypes_int64(Dl), ctypes_double(w_bm25), ctypes_double(w_sim),
               L.ptr(max_a), L.ptr(max_b), L.ptr(final), self.device, L.current_stream_ptr())
        kk = min(k, Dl)
        ids = np.empty((nq, kk), dtype=np.int32)
        vals = np.empty((nq, kk), dtype=np.float64)
        L.call("hipts_topk", L.ptr(final), nq, ctypes_int64(Dl), kk, L.ptr(ids), L.ptr(vals), L.HOST, self.device, L.current_stream_ptr())
        return vals, ids.astype(np.int64) + self.lo

    def score_topk(self, query_weights: Sequence[dict], query_vectors: np.ndarray, k: int, dist=None):
        """The collective form: every rank calls it with the same queries; every rank returns the merged
        (ids int64 [nq,k], scores float64 [nq,k])."""
        import torch
        bm, sims, max_a, max_b = self.local_scores(query_weights, query_vectors)
        multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        # RCCL ("nccl") moves device tensors; the gloo backend of the CPU / single-GPU tests moves host tensors
        cdev = bm.device if multi and dist.get_backend() == "nccl" else torch.device("cpu")
        if multi:
            ga, gb = max_a.to(cdev), max_b.to(cdev)
            dist.all_reduce(ga, op=dist.ReduceOp.MAX)
            dist.all_reduce(gb, op=dist.ReduceOp.MAX)
            max_a, max_b = ga.to(bm.device), gb.to(bm.device)
        vals, ids = self.local_topk(bm, sims, max_a, max_b, k)
        if not multi:
            return merge_topk([vals], [ids], k)
        # candidates: pad to k columns with (-inf, sentinel id) so that all ranks send equal shapes
        nq = vals.shape[0]
        pv = np.full((nq, k), -np.inf, dtype=np.float64)
        pi = np.full((nq, k), np.iinfo(np.int64).max, dtype=np.int64)
        pv[:, :vals.shape[1]], pi[:, :ids.shape[1]] = vals, ids
        tv, ti = torch.from_numpy(pv).to(cdev), torch.from_numpy(pi).to(cdev)
        world = dist.get_world_size()
        gv = torch.empty((world * nq, k), dtype=tv.dtype, device=cdev)          # rank-major concatenation
        gi = torch.empty((world * nq, k), dtype=ti.dtype, device=cdev)
        dist.all_gather_into_tensor(gv, tv.contiguous())
        dist.all_gather_into_tensor(gi, ti.contiguous())
        gv, gi = gv.cpu().numpy().reshape(world, nq, k), gi.cpu().numpy().reshape(world, nq, k)
        ids_m, vals_m = merge_topk(list(gv), list(gi), k)
        keep = min(k, self.D)
        return ids_m[:, :keep], vals_m[:, :keep]


def ctypes_double(x):
    import ctypes
    return ctypes.c_double(float(x))


def ctypes_int64(x):
    import ctypes
    return ctypes.c_int64(int(x))
