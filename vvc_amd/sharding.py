"""Frame sharding for multi-GPU runs: one process per GPU, frames are independent units, so every rank takes a
contiguous range of global frame indices and there is no data-path collective.  torch.distributed is used only for
the barrier around timed regions and for max-over-ranks of the measured time (RCCL on GPUs, gloo in CPU tests)."""
import os

import torch


def frame_range(rank, world, total_frames):
    """Global frame indices [lo, hi) of `rank`: contiguous ranges [g*N/G, (g+1)*N/G)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return rank * total_frames // world, (rank + 1) * total_frames // world


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, force=False):
    """Join the process group described by the torchrun environment.  Returns (rank, world, local_rank, dist|None).
    A world of one rank needs no group unless force is set (used to rehearse the RCCL path on a single GPU)."""
    rank, world, local_rank = env_rank()
    if world == 1 and not force:
        return rank, world, local_rank, None
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, **kw)
    return rank, world, local_rank, dist


def barrier(dist):
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(dist, values):
    """Element-wise maximum of a list of floats over all ranks."""
    if dist is None:
        return list(values)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def gather_floats(dist, value):
    """One float per rank, in rank order, on every rank (host-side, tiny)."""
    if dist is None:
        return [float(value)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, float(value))
    return parts


def sum_over_ranks_u64(dist, value):
    """Sum modulo 2^64 of one unsigned 64-bit value per rank (checksum of checksums; host-side, tiny)."""
    if dist is None:
        return value & 0xFFFFFFFFFFFFFFFF
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, int(value))
    return sum(parts) & 0xFFFFFFFFFFFFFFFF
