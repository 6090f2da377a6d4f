"""One-subject-per-GPU batch mode.

One subject's multi-channel reconstruction stays on one GPU: channels couple
through the joint-TV z-update and the pull/push footprints are non-local, so
nothing shards inside a subject ("replicas only").  A batch of independent
subjects is embarrassingly parallel: rank g of G takes subjects g, g+G, ...; no
volume data ever crosses xGMI.  torch.distributed (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests) is used only for the start/end barriers and for
gathering a few scalars per rank.
"""
import os
import time

import torch
import torch.distributed as dist

from ._host import configure_host

host_config = None  # what init_from_env applied: dict(threads, cpus)


def partition(n_subjects, world_size, rank):
    """Indices of the subjects rank ``rank`` reconstructs (round-robin)."""
    if not (0 <= rank < world_size):
        raise ValueError('rank out of range')
    return list(range(rank, n_subjects, world_size))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun style).
    Returns (rank, world_size, local_rank).  world_size 1 needs no process group.

    Also gives the rank its share of the host (``_host.configure_host``): torch intra-op threads capped at
    cores / local world size (<= 8), the process pinned to a core set of its own - NUMA-local to its GPU
    where sysfs says which node that is.  LOCAL_WORLD_SIZE (torchrun) counts the ranks of this host."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    global host_config
    host_config = configure_host(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def _sync(device):
    if device is not None and torch.device(device).type == 'cuda':
        torch.cuda.synchronize(device)


def run_batch(n_subjects, reconstruct, device=None):
    """Run ``reconstruct(subject_index) -> float`` (e.g. a final objective value) for
    this rank's share of the batch, bracketed by barriers.

    Returns a dict on every rank: ``results`` {subject: value} for ALL subjects,
    ``elapsed`` = max over ranks of the wall time between the barriers, and
    ``subjects_per_sec`` = n_subjects / elapsed (whole job)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = partition(n_subjects, world, rank)
    _sync(device)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    local = {s: float(reconstruct(s)) for s in mine}
    _sync(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    results = dict(local)
    if world > 1:
        dev = device if (device is not None and dist.get_backend() == 'nccl') else 'cpu'
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        results = {}
        for part in gathered:
            results.update(part)
    return dict(results=results, elapsed=elapsed, subjects_per_sec=n_subjects / elapsed,
                world_size=world, rank=rank, mine=mine)
