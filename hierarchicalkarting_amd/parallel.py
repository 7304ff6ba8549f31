"""Multi-GPU: race instances are independent (one RacingEnvController owns its own Agents[] / Sections[], REC:46-52), so
envs shard contiguously over ranks with NO data-path collective.  The path's only exchange step is the gather of the
per-episode results (32 B per agent) — one all-gather over RCCL/xGMI (backend "nccl" on ROCm), or gloo on CPU."""
import numpy as np
from .env import RESULT_DT


def shard_range(total_envs, rank, world):
    """contiguous env-id range [lo, hi) owned by `rank`"""
    per = (total_envs + world - 1) // world
    lo = min(rank * per, total_envs)
    return lo, min(lo + per, total_envs)


def gather_episode_results(env_or_array, dist=None):
    """-> structured array [E_total][A] on every rank.  `env_or_array`: a RacingEnv (device results) or a local
    numpy RESULT_DT array (used by the CPU gloo tests)."""
    local = env_or_array if isinstance(env_or_array, np.ndarray) else env_or_array.episode_results()
    if dist is None or not dist.is_initialized():
        return local
    import torch
    world = dist.get_world_size()
    backend = dist.get_backend()
    raw = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy())
    if backend == "nccl":
        raw = raw.cuda()
    out = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(out, raw)
    parts = [o.cpu().numpy().view(RESULT_DT).reshape(local.shape) for o in out]
    return np.concatenate(parts, axis=0)
