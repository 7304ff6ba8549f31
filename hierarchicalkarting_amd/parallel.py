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
    is_env = not isinstance(env_or_array, np.ndarray)
    if dist is None or not dist.is_initialized():
        return env_or_array.episode_results() if is_env else env_or_array
    import torch
    world = dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    if is_env and dev == "cuda":
        # RCCL: the collective reads the library's own device buffer (a zero-copy view, RacingEnv.results_tensor) — no device -> host -> device bounce
        raw = env_or_array.results_tensor().reshape(-1)
        A = env_or_array.A
    else:
        local = env_or_array.episode_results() if is_env else env_or_array
        A = local.shape[1]
        raw = torch.from_numpy(np.ascontiguousarray(local).view(np.uint8).reshape(-1).copy()).to(dev)
    # ranks may hold different env counts (shard_range of a total the world size does not divide): exchange the byte
    # counts, pad every contribution to the largest (all_gather needs equal sizes), trim after the gather
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([raw.numel()], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    per = max(sizes)
    if raw.numel() < per:
        raw = torch.cat([raw, torch.zeros(per - raw.numel(), dtype=torch.uint8, device=dev)])
    out = [torch.empty(per, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(out, raw)
    parts = [o[:n].cpu().numpy().view(RESULT_DT).reshape(-1, A) for o, n in zip(out, sizes)]
    return np.concatenate(parts, axis=0)
