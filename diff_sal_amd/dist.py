"""One-process-per-GPU plumbing for the sampling path (torch.distributed; backend "nccl" is RCCL on ROCm).

Inference shards BY CLIP: every rank holds a full replica of the denoiser and samples its own contiguous
block of clips; there is no collective on the data path ("replicas only").  Collectives are used
only around it: a barrier before/after timing, a MAX-reduce of elapsed time, and an optional final gather
of the predictions.  The reference does the same sharding through DistributedSampler and gathers nothing
(R/datasets/prepare_data.py:13-32, R/diffusion_trainer.py:898-935).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the default group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced block of [0, n_items) owned by ``rank`` (first n_items % world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sample_sharded(sampler, x_T: torch.Tensor, feats: List[torch.Tensor], audio: Optional[torch.Tensor],
                   batch: int, gather: bool = False):
    """Sample ``x_T.shape[0]`` clips split over the ranks of the default group, ``batch`` clips at a time.

    Every argument holds ALL clips on every rank (host or device); each rank only touches its shard.
    Returns this rank's predictions, or (gather=True) the full tensor in clip order on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_range(x_T.shape[0], rank, world)
    outs = []
    for s in range(mine.start, mine.stop, batch):
        e = min(s + batch, mine.stop)
        a = None if audio is None else audio[s:e]
        fs = [f[s:e] for f in feats]
        if sampler.sample_type == "ddim":          # same dispatch as DiffusionSampler.sample_image
            outs.append(sampler.sample_ddim(x_T[s:e], fs, a))
        elif sampler.sample_type in ("dpmsolver", "dpmsolver++"):
            outs.append(sampler.sample_dpm_solver(x_T[s:e], fs, a))
        elif sampler.sample_type == "ddpm":
            outs.append(sampler.sample_ddpm(x_T[s:e], fs, a))
        else:
            raise NotImplementedError(sampler.sample_type)
    local = torch.cat(outs) if outs else x_T[:0]
    if not gather or world == 1:
        return local
    sizes = [len(shard_range(x_T.shape[0], r, world)) for r in range(world)]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)])
