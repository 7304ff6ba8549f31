"""Sharding invariance: an env's trajectory depends on its GLOBAL id only — start jitter, planner draws, actor
sampling and Training-mode scatter are all keyed by env_id_base + env — so E envs on one handle and the same E envs split
over two handles (as two ranks would hold them) give identical records."""
import numpy as np
import pytest
from hierarchicalkarting_amd import _lib

pytestmark = pytest.mark.gpu


def test_two_shards_equal_one_handle():
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd.policy import Policy
    kw = dict(num_agents=4, jitter_seed=0x5EED0000, rewards=1, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 0, 0, 1], laps=1,
              max_episode_steps=260, mcts_iterations=10, tree_search_depth=[5, 8, 5, 5],
              high_mode=[_lib.HK_HIGH_FIXED, _lib.HK_HIGH_MCTS, _lib.HK_HIGH_FIXED, _lib.HK_HIGH_FIXED],
              low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_LQR, _lib.HK_LOW_LQR, _lib.HK_LOW_RL])

    def run(E, base):
        g = hk.RacingEnv(hk.make_config(E, env_id_base=base, **kw))
        pol = Policy.random(g.obs_dim * 4, 64, 2, seed=21)
        g.attach_policy(pol, [0, 3], 2)
        g.reset()
        g.step(123); g.step(300)
        return g.agent_state(), g.episode_results(), g.mcts_state()

    whole = run(12, 0)
    lo, hi = run(6, 0), run(6, 6)
    for w, a, b in zip(whole, lo, hi):
        both = np.concatenate([a, b], axis=0)
        assert w.tobytes() == both.tobytes()
    assert (whole[1]["episode"] >= 0).all()
