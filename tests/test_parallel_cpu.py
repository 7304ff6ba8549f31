"""The N > 1 path on CPU: two gloo ranks, envs sharded contiguously (env_id_base), no data-path collective, one
all-gather of the episode results.  The union of the shards must equal a single-process run over all envs."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, A, TICKS = 7, 2, 260       # 7 envs over 2 ranks: shards of 4 and 3 (the gather pads and trims)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from hierarchicalkarting_amd.config import make_config
    from hierarchicalkarting_amd.parallel import shard_range, gather_episode_results
    lo, hi = shard_range(TOTAL, rank, world)
    b = make_config(hi - lo, A, jitter_seed=0x5EED0000, env_id_base=lo, max_episode_steps=250)
    o = O.OracleEnv(b)
    o.reset()
    o.step(TICKS)                                   # past the 250-tick timeout: every env has one finished episode
    allres = gather_episode_results(o.episode_results(), dist)
    px = o.agent_state()["px"]
    q.put((rank, lo, hi, allres.tobytes(), allres.shape, px.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from hierarchicalkarting_amd.config import make_config
    from hierarchicalkarting_amd.env import RESULT_DT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    b = make_config(TOTAL, A, jitter_seed=0x5EED0000, env_id_base=0, max_episode_steps=250)
    o = O.OracleEnv(b)
    o.reset(); o.step(TICKS)
    ref = o.episode_results()
    refpx = o.agent_state()["px"]
    for rank, lo, hi, raw, shape, px in got:
        allres = np.frombuffer(raw, RESULT_DT).reshape(shape)
        assert allres.shape == ref.shape
        for name in ref.dtype.names:
            assert np.array_equal(allres[name], ref[name]), name          # every rank holds the full gathered table
        assert np.array_equal(np.frombuffer(px, np.float32).reshape(hi - lo, A), refpx[lo:hi])
    assert (ref["episode"] == 0).all()


def test_shard_range():
    from hierarchicalkarting_amd.parallel import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_range(65536 * 8, 7, 8) == (65536 * 7, 65536 * 8)
    assert shard_range(2, 3, 4) == (2, 2)
