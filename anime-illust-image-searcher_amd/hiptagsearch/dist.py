"""One process per GPU: process-group set-up for the CLIs.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tagging.py --dir D

`init_from_env()` must run BEFORE anything touches the GPU (it only counts devices, which does not initialise HIP on this
image).  Backend "nccl" is RCCL over xGMI, one GPU per rank; HIPTS_DIST_BACKEND=gloo lets several ranks share one GPU and
moves host tensors instead -- the rehearsal mode of the one-GPU tests.  The reference has no multi-device path
(SURVEY.md section 2.3): the work it does in one loop (tagging.py:276-359, gen_cfeatures.py:337-459) is cut into contiguous
blocks of the file list here, and ONE all-gather of fixed-width rows puts the results back in file order."""
import os
from typing import Tuple


def init_from_env(device_arg: int = 0):
    """Returns (dist module or None, rank, world, device index for this rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None, 0, 1, device_arg
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("HIPTS_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl":
        if local >= ndev:
            raise SystemExit("rank %d: LOCAL_RANK %d but %d GPU(s) visible (HIPTS_DIST_BACKEND=gloo shares one GPU between ranks)"
                             % (rank, local, ndev))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        device = local
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
        device = local % max(ndev, 1)
    torch.cuda.set_device(device)
    return dist, rank, world, device


def broadcast_object(obj, dist, src: int = 0):
    """Rank `src`'s Python object on every rank (file lists: every rank must cut the SAME list)."""
    if dist is None:
        return obj
    box = [obj if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def collective_device(dist, device: int):
    """Where tensors handed to a collective must live: the rank's GPU under RCCL, the host under gloo."""
    import torch
    return torch.device("cuda", device) if dist is not None and dist.get_backend() == "nccl" else torch.device("cpu")


def finish(dist):
    if dist is not None and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
