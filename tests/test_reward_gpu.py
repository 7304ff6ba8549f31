"""Reward shaping inside the fused tick kernel vs the CPU oracle: cumulative / per-decision / group accumulators of every
agent bit-identical tick by tick (float adds replayed in the same order), terminal group rewards in the episode results,
hk_get_rewards read-and-reset semantics."""
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib

pytestmark = pytest.mark.gpu


def _pair(E, A, **kw):
    import hierarchicalkarting_amd as hk
    b = hk.make_config(E, A, rewards=1, **kw)
    g = hk.RacingEnv(b)
    o = O.OracleEnv(b)
    g.reset(); o.reset()
    return g, o


def _cmp(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name, np.argwhere(x != y)[:3].tolist())
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "episodes_done", "status"):
        assert np.array_equal(ge[name], oe[name]), (t, name)


def test_two_agents_tick_by_tick():
    g, o = _pair(6, 2, jitter_seed=4)
    for t in range(1, 401):
        g.step(1); o.step(1)
        _cmp(g, o, t)
    a = g.agent_state()
    assert (a["cum_reward"] > 20).all() and (a["group_reward"] > 20).all()


def test_four_agents_2v2_with_read_and_reset():
    g, o = _pair(32, 4, jitter_seed=9, training_agents=[1, 1, 1, 1])
    t = 0
    for n in (75, 3, 100, 57, 200, 65, 300):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
        gr, gg = g.rewards(); orr, og = o.rewards()
        assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32)) and np.array_equal(gg.view(np.uint32), og.view(np.uint32)), t
        a = g.agent_state()
        assert (a["step_reward"] == 0).all() and (a["group_reward"] == 0).all()      # Agent.SendInfo zeroes them


def test_episode_end_goal_timing_and_resets():
    """1-lap races with a tight time-out: finishes, time-outs, goal-timing rewards (disableOnEnd off so they reach the
    groups), table resets, next episode"""
    g, o = _pair(24, 4, jitter_seed=2, laps=1, max_episode_steps=1200, training_agents=[1, 0, 1, 1], disable_on_end=0)
    t = 0
    for n in (400, 400, 200, 150, 100, 250, 500):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
    gr, orr = g.episode_results(), o.episode_results()
    for name in gr.dtype.names:
        x, y = gr[name], orr[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), name
    assert (gr["episode"] >= 0).all() and (gr["group_reward"] != 0).any() and (gr["reward"] != 0).all()


def test_rl_and_mcts_agents_with_rewards():
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd.policy import Policy
    b = hk.make_config(8, 4, rewards=1, jitter_seed=6, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_LQR, _lib.HK_LOW_LQR, _lib.HK_LOW_LQR],
                       high_mode=[_lib.HK_HIGH_FIXED, _lib.HK_HIGH_MCTS, _lib.HK_HIGH_FIXED, _lib.HK_HIGH_MCTS],
                       tree_search_depth=[5, 8, 5, 8], mcts_iterations=12)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=3)
    g.attach_policy(pol, [0], 2); o.attach_policy(pol, [0], 2)
    t = 0
    for n in (80, 45, 100, 75):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)


def test_hit_penalties_from_observations():
    """HKA:580-598: karts parked against a wall and behind each other; hk_get_observations raises the events"""
    g, o = _pair(3, 4, jitter_seed=0, jitter_pos=0.0, jitter_yaw=0.0)
    for e in (g, o):
        e.step(80); e.rewards()
        st = e.agent_state()
        for k in range(3):
            st["px"][k] = [19.6 - 0.05 * k, 15.0, 15.0, 15.9]; st["pz"][k] = [2.0, 30.0, 31.3, 30.6]
            st["yaw"][k] = [np.pi / 2, 0.0, 0.0, 3.3]
        st["vx"][:] = 0; st["vz"][:] = 0
        e.set_agent_state(st)
    assert np.array_equal(g.observations(), o.observations())
    _cmp(g, o, 80)
    gr, gg = g.rewards(); orr, og = o.rewards()
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32)) and (gr < -0.04).any()
    # and through the decision loop of an attached policy
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd.policy import Policy
    b = hk.make_config(16, 2, rewards=1, jitter_seed=3, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_RL])
    g = hk.RacingEnv(b); o = O.OracleEnv(b); g.reset(); o.reset()
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=8)
    g.attach_policy(pol, [0, 1], 2); o.attach_policy(pol, [0, 1], 2)
    for t in (150, 151, 200, 97):
        g.step(t); o.step(t)
        _cmp(g, o, t)
