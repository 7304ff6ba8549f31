"""hk_comm_* / hk_gather_results: the native RCCL all-gather of the episode results.  One GPU is what the test box has, so
this runs the world_size = 1 communicator (bootstrap, all-gather, layout); the multi-rank path is the same call with more
ranks and is what bench.py's torch.distributed gather (gloo-tested on CPU) does at the Python level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_gather_results_single_rank():
    import hierarchicalkarting_amd as hk
    g = hk.RacingEnv(hk.make_config(9, 2, jitter_seed=3, laps=1, max_episode_steps=900))
    g.reset(); g.step(950)
    cid = hk.RacingEnv.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    g.comm_init(1, 0, cid)
    allr = g.gather_results()
    loc = g.episode_results()
    assert allr.shape == (9, 2) and allr.tobytes() == loc.tobytes() and (allr["episode"] >= 0).all()
    with pytest.raises(hk.HkError):
        g.comm_init(1, 0, cid)                       # one communicator per handle
    g.comm_destroy()
    with pytest.raises(hk.HkError):
        g.gather_results()                           # no communicator any more
    g.comm_init(1, 0, hk.RacingEnv.comm_unique_id()) # a new one can be made
    assert g.gather_results().tobytes() == loc.tobytes()
