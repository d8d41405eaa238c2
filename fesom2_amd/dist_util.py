"""torch.distributed plumbing for bench.py: one process per GPU, RCCL (backend "nccl") on the GPU node, gloo on CPU.
Only the barrier / max-over-ranks timing contract lives here; the data path has no collective in this round
(N>1 = independent replicas)."""
import os
import time


def init(backend):
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def timed_region(fn, world, sync=None, device="cpu"):
    """barrier + sync, run fn, sync + barrier; returns the MAX elapsed seconds over ranks."""
    import torch
    if sync:
        sync()
    barrier(world)
    t0 = time.perf_counter()
    fn()
    if sync:
        sync()
    barrier(world)
    el = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def aggregate_sypd(seconds_per_step, world, steps_per_year):
    """replicas: every rank advances its own copy of the mesh -> ensemble simulated-years/day"""
    return world * 86400.0 / (steps_per_year * seconds_per_step)
