"""Multi-GPU sharding of independent environments.

Envs never interact, so the data path has no collective: each rank owns a
contiguous block of envs on its own GPU (one process per GPU, RCCL only for the
optional trajectory gather).  `gather_trajectories` is the single collective the
north-star names: an all-gather of the per-step trajectory slab.
"""

from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK (torchrun env).
    Returns (rank, world_size, local_rank).  No-op for world size 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def shard_envs(total_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of the global env index range owned by `rank`.
    Blocks differ by at most one env; every env has exactly one owner."""
    if total_envs < 0 or not 0 <= rank < world_size:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total_envs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """Independent RNG stream per rank (seed + 1000*rank)."""
    return int(seed) + 1000 * int(rank)


def pack_trajectory_record(qpos, reward, discount, step_type, key_activation, dtype=None):
    """Compact per-env record (§8e): qpos | reward | discount | step_type | activation bits, one row per env.
    `dtype`: torch.float32 (half the xGMI bytes; qpos rounded to single) or torch.float64 (the engine's
    precision survives the gather); default = the dtype of `qpos`.  The activation bits travel as raw bytes
    in the last 3 (float32) / 2 (float64) columns."""
    dtype = dtype or (qpos.dtype if qpos.dtype in (torch.float32, torch.float64) else torch.float32)
    E, nk = qpos.shape[0], key_activation.shape[1]
    nbytes = 12 if dtype == torch.float32 else 16
    # bits -> bytes (little-endian inside each word), a handful of launches
    byte_w = (1 << torch.arange(8, device=key_activation.device, dtype=torch.int32))
    padded = torch.zeros((E, 8 * nbytes), dtype=torch.int32, device=key_activation.device)
    padded[:, :nk] = key_activation
    by = (padded.view(E, nbytes, 8) * byte_w).sum(2).to(torch.uint8)        # [E, nbytes]
    bits = by.contiguous().view(dtype)                                        # [E, 3] or [E, 2]
    return torch.cat([qpos.to(dtype), reward.to(dtype).reshape(E, 1), discount.to(dtype).reshape(E, 1),
                      step_type.to(dtype).reshape(E, 1), bits], dim=1).contiguous()


def unpack_key_activation(record: torch.Tensor, nv: int, n_keys: int = 88) -> torch.Tensor:
    raw = record[:, nv + 3:].contiguous().view(torch.uint8).to(torch.int64)    # [E, 12 or 16] bytes
    shifts = torch.arange(8, device=record.device, dtype=torch.int64)
    bits = ((raw[:, :, None] >> shifts) & 1).reshape(record.shape[0], -1)
    return bits[:, :n_keys].bool()


def gather_trajectories(local: torch.Tensor, async_op: bool = False, out: torch.Tensor | None = None,
                        force_collective: bool = False):
    """All-gathers equally sized per-rank slabs [E_local, ...] into [world*E_local, ...]
    ordered by rank (== global env order under `shard_envs`).  With `async_op` the
    collective is only enqueued (it overlaps the next env step) and `(out, work)` is
    returned; call `work.wait()` before reading `out`.  A world of one rank returns `local` itself unless
    `force_collective` (the single-GPU RCCL self-test: the collective call then really goes through the backend)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return (local, None) if async_op else local
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
    work = dist.all_gather_into_tensor(out, local.contiguous(), async_op=async_op)
    return (out, work) if async_op else out
