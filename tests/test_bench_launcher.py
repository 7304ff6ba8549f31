"""bench.py --gpus N must produce N ranks by itself: the parent spawns the ranks before anything touches
the GPU, rank 0 prints one JSON line with n_gpus = N, and a box with fewer GPUs gets a clean error instead of a silent
1-GPU run.  The rank plumbing (rendezvous, unequal shards, barrier, max over ranks, padded all-gather) runs here on
CPU with gloo through --selftest-launcher; no libhk compute is involved."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=600)


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    p = _run(["--gpus", "2", "--selftest-launcher", "--envs-per-gpu", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gather_ok"] and out["gathered_envs"] == 11      # 6 + 5: unequal shards
    assert abs(out["max_time"] - 0.002) < 1e-12                                        # max over ranks
    assert out["ranks_seen"] == 2 and [round(t, 9) for t in out["per_rank_ms"]] == [1.0, 2.0]      # every rank's own span, rank order


def test_gpus_2_without_two_gpus_fails_cleanly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has 2 GPUs")
    p = _run(["--gpus", "2"])
    assert p.returncode == 2
    assert "needs 2 GPUs" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]                 # no silent n_gpus = 1 line


def test_gpus_flag_must_match_world_size():
    p = _run(["--gpus", "2", "--selftest-launcher"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
