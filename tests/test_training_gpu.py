"""Training mode on device (random scatter resets inside the tick kernel and in hk_reset, planRandomly) vs the CPU oracle."""
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib

pytestmark = pytest.mark.gpu
TR = _lib.HK_MODE_TRAINING


def _cmp(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name, np.argwhere(x != y)[:3].tolist())
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "episodes_done", "experiment_num", "status"):
        assert np.array_equal(ge[name], oe[name]), (t, name)


def _pair(E, A, **kw):
    import hierarchicalkarting_amd as hk
    b = hk.make_config(E, A, env_mode=TR, **kw)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    return g, o


def test_scatter_and_random_plans_match():
    g, o = _pair(300, 4, training_agents=[1, 1, 1, 1], laps=2, jitter_seed=0)
    _cmp(g, o, 0)
    g.reset([5, 17, 100], 3); o.reset([5, 17, 100], 3)
    _cmp(g, o, 0)


def test_training_episodes_with_timeouts_rewards_and_complex_track():
    g, o = _pair(24, 4, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=1, jitter_seed=0, track="complex")
    t = 0
    for n in (100, 1, 199, 57, 243, 300):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
    assert (g.env_state()["episodes_done"] >= 2).all()


def test_training_with_mcts_and_policy():
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd.policy import Policy
    b = hk.make_config(12, 2, env_mode=TR, training_agents=[1, 0], rewards=1, high_mode=[_lib.HK_HIGH_MCTS, _lib.HK_HIGH_MCTS],
                       low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_LQR], tree_search_depth=8, mcts_iterations=10, laps=1,
                       max_episode_steps=350, jitter_seed=0)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=4)
    g.attach_policy(pol, [0], 2); o.attach_policy(pol, [0], 2)
    g.reset(); o.reset()
    t = 0
    for n in (120, 130, 101, 99, 250):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
