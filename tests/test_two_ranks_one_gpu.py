"""The N > 1 product path with the real library: two PROCESSES, each with its own libhk handle on the
one GPU of the box, shards of unequal size (env_id_base), gloo for rendezvous / barrier / max / gather.  The union of the two
shards — agent records, env words and the gathered episode results — must equal one handle over all envs.  RCCL itself needs
one device per rank and cannot run with two ranks on one device; everything else bench.py's rank code does is exercised here:
the second case runs `bench.py --gpus 2 --same-device` (its rank code with gloo in place of nccl) and checks the JSON line."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, A, TICKS = 4101, 4, 300            # shards of 2051 and 2050 envs; past the 250-tick timeout: one finished episode each

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd.parallel import shard_range, gather_episode_results
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard_range(%(total)d, rank, world)
env = hk.RacingEnv(hk.make_config(hi - lo, %(A)d, jitter_seed=0x5EED0000, env_id_base=lo, max_episode_steps=250, device_id=0))
env.reset()
dist.barrier()
for _ in range(%(ticks)d // 100):
    env.step(100)                      # both processes drive the same GPU at once
env.synchronize()
allres = gather_episode_results(env, dist)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), lo=lo, hi=hi, agents=env.agent_state(), envs=env.env_state(), allres=allres)
dist.barrier()
dist.destroy_process_group()
'''


def _launch(args, env_extra=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_two_processes_on_one_gpu_equal_one_handle(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "total": TOTAL, "A": A, "ticks": TICKS, "out": str(tmp_path)})
    p = _launch([str(script)])
    assert p.returncode == 0, p.stderr[-2000:]
    import hierarchicalkarting_amd as hk
    one = hk.RacingEnv(hk.make_config(TOTAL, A, jitter_seed=0x5EED0000, env_id_base=0, max_episode_steps=250))
    one.reset()
    for _ in range(TICKS // 100):
        one.step(100)
    ref_a, ref_e, ref_r = one.agent_state(), one.env_state(), one.episode_results()
    assert (ref_r["episode"] == 0).all()
    sizes = []
    for rank in range(2):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        lo, hi = int(d["lo"]), int(d["hi"])
        sizes.append(hi - lo)
        for name in ref_a.dtype.names:
            assert np.array_equal(d["agents"][name], ref_a[name][lo:hi]), (rank, name)
        for name in ("episode_steps", "inactive_mask", "experiment_num", "episodes_done", "status"):
            assert np.array_equal(d["envs"][name], ref_e[name][lo:hi]), (rank, name)
        assert d["allres"].shape == ref_r.shape                      # every rank holds the full gathered table
        for name in ref_r.dtype.names:
            assert np.array_equal(d["allres"][name], ref_r[name]), (rank, name)
    assert sizes == [2051, 2050]


def test_bench_rank_code_with_two_ranks_on_one_gpu():
    p = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--envs-per-gpu", "4096", "--steps", "40", "--warmup", "4",
                 "--preroll", "100", "--no-secondary", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1                                              # rank 0 prints ONE line
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 40 and out["value"] > 0
    assert out["config"]["envs_per_gpu"] == 4096 and out["config"]["same_device"] is True
    # value = the units ALL ranks processed / the max-over-ranks time
    assert abs(out["value"] - 2 * 4096 * 40 / (out["ms_per_step"] * 1e-3 * 40)) / out["value"] < 1e-6
    # round 6: every rank's own span is in the line, the time is their maximum (no barrier inside a span), the gather has its own clock
    mg = out["multi_gpu"]
    assert mg["ranks_seen"] == 2 and len(mg["per_rank_ms"]) == 2 and all(t > 0 for t in mg["per_rank_ms"])
    assert abs(max(mg["per_rank_ms"]) - out["ms_per_step"] * 40) / max(mg["per_rank_ms"]) < 1e-6
    assert mg["result_gather"]["envs_gathered"] == 2 * 4096 and mg["result_gather"]["ms"] > 0


def test_bench_rank_code_over_rccl_with_one_rank():
    """the RCCL leg of bench.py's rank code (process group on `nccl`, per-rank spans gathered as CUDA tensors, the result gather straight from the
    library's device buffer) with the one rank a 1-GPU box allows: HK_BENCH_FORCE_DIST=1 makes a world of 1 go through it"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HK_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + (os.getpid() % 2000) + 7),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs-per-gpu", "4096", "--steps", "40", "--warmup", "4", "--preroll", "300",
                        "--no-secondary", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1
    out = json.loads(line[0])
    mg = out["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["ranks_seen"] == 1 and len(mg["per_rank_ms"]) == 1
    assert abs(mg["per_rank_ms"][0] - out["ms_per_step"] * 40) / mg["per_rank_ms"][0] < 1e-6
    assert mg["result_gather"]["envs_gathered"] == 4096 and out["config"]["finished_episodes_seen"] >= 0
    assert out["config"]["build"]["units"]["hk_ga4.hip"]["guard_findings"] == 0 and out["config"]["schedule"]["call_ticks"] == 40
