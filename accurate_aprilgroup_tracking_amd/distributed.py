"""Multi-GPU layout: streams are independent, so they shard across ranks with no data-path
collective.  One process per GPU; rank r owns streams
[r*S/W, (r+1)*S/W).  The only exchange is the gather of the 6-DoF poses, once per chunk of
frames (64 B per stream-frame -> latency-bound; never per frame): one all_gather over
RCCL/xGMI (backend "nccl" on ROCm), or gloo in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, force=False):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  A single process (world 1) joins
    nothing unless `force` (the world-1 RCCL test: process group of one rank, so that the collective branch of gather_poses
    runs on a machine with one GPU)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("AGT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_streams(n_streams, rank, world):
    """Contiguous block of stream indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_streams, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def gather_poses(local, world=None, force_collective=False):
    """local: [frames, streams_local, 8] f64 (rvec 3, tvec 3, ok, frame index).  Returns
    [world, frames, streams_local, 8] on every rank (equal shard sizes required), or
    local[None] when not distributed.  force_collective: take the all_gather branch even for a process group of ONE rank
    (tests/test_gpu_distributed.py: RCCL initialisation + a device-tensor all_gather on the one GPU a test box has)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return local.unsqueeze(0)
    world = dist.get_world_size()
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the multi-rank flow without RCCL (several ranks sharing one GPU): gather on the host
        host = local.cpu()
        out = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host)
        return out.view((world,) + tuple(host.shape)).to(local.device)
    # concatenated along dim 0 (the layout both RCCL and gloo accept), viewed as [world, ...]
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out.view((world,) + tuple(local.shape))


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def backend_name():
    """"nccl" is RCCL on ROCm; "none" when the job is a single process"""
    if not dist.is_initialized():
        return "none"
    b = dist.get_backend()
    return "rccl (torch.distributed nccl)" if b == "nccl" else str(b)
