"""The synthetic 8-agent configuration (BASELINE configs[4]: 8-agent Complex, mixed MCTS-RL vs MCTS-LQNG; the reference has no
8-agent scene): the 8-lane-per-env kernels against the oracle, field for field — Fixed-LQNG with games of up to
8 players, the planner with 8 karts in the discrete game, the RL actor on 126-float observations, rewards, Training mode."""
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib

pytestmark = pytest.mark.gpu


def _cmp(g, o, t, obs=True):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name, np.argwhere(x != y)[:4])
    if obs:
        assert np.array_equal(g.observations().view(np.uint32), o.observations().view(np.uint32)), t
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "experiment_num", "episodes_done"):
        assert np.array_equal(ge[name], oe[name]), (t, name)


@pytest.mark.parametrize("track", ["oval", "complex"])
def test_fixed_lqng_8_agents(track):
    import hierarchicalkarting_amd as hk
    b = hk.make_config(24, 8, track=track, jitter_seed=0x5EED0000, laps=1)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    assert g.observations().shape[-1] == 126
    t = 0
    for n in (76, 4, 40, 80, 100, 100, 200, 400, 500):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
    assert np.array_equal(g.episode_results(), o.episode_results())


def _pack(st, track):
    """all 8 karts of every env packed behind the first Trigger: two rows of four lanes, 2.6 m apart — within 8 m of each other"""
    s0 = track["sections"][0]
    st = st.copy()
    for j in range(8):
        lane = j % 4 + 1
        st["px"][:, j] = s0["Lane%d" % lane]["x"] + 0.0
        st["pz"][:, j] = s0["Lane%d" % lane]["z"] + 2.0 + 2.6 * (j // 4)
        st["lane"][:, j] = lane
        st["section_index"][:, j] = 0
        st["init_checkpoint_index"][:, j] = 0
    return st


def test_big_games_8_agents():
    """games with 5..8 players (lqn_big_kernel): the whole field packed within 8 m, free-for-all and 4v4"""
    import os
    import hierarchicalkarting_amd as hk
    ffa = (list(range(8)), [[] for _ in range(8)], [[j for j in range(8) if j != i] for i in range(8)])
    for wiring in (ffa, None):
        b = hk.make_config(12, 8, jitter_seed=3, wiring=wiring, laps=1, max_episode_steps=1500)
        os.environ["HK_LQ_DEBUG"] = "1"
        try:
            g = hk.RacingEnv(b); o = O.OracleEnv(b)
        finally:
            del os.environ["HK_LQ_DEBUG"]
        g.reset(); o.reset()
        st = _pack(o.agent_state(), b.track)
        # yaw of section 0 points along +z on both tracks? keep the reset yaw (jittered) — only the positions move
        g.set_agent_state(st); o.set_agent_state(st)
        seen = set()
        t = 0
        for n in (76, 4, 4, 4, 4, 8, 20, 40, 80, 160):
            g.step(n); o.step(n); t += n
            _cmp(g, o, t)
            for env in range(0, 12, 5):
                for ego in range(8):
                    seen.add(int(o.lq_debug(env, ego).n_players))
                    gd, od = g.lq_debug(env, ego), o.lq_debug(env, ego)
                    assert gd.n_players == od.n_players and gd.u0[0] == od.u0[0] and gd.u0[1] == od.u0[1], (t, env, ego)
        assert max(seen) > 6, seen


def test_mcts_rl_rewards_8_agents():
    import hierarchicalkarting_amd as hk
    A = 8
    low = [_lib.HK_LOW_RL] * 4 + [_lib.HK_LOW_LQR] * 4                 # "mixed MCTS-RL vs MCTS-LQNG"
    b = hk.make_config(12, A, track="complex", jitter_seed=11, high_mode=_lib.HK_HIGH_MCTS, low_mode=low, tree_search_depth=3,
                       mcts_iterations=16, rewards=1, training_agents=[0] * A, laps=1, section_window=3)
    pol = hk.Policy.random(126 * 4, 128, 2, seed=5, stack=4)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.attach_policy(pol, [0, 1, 2, 3], 2)
    o.attach_policy(pol, [0, 1, 2, 3], 2)
    g.reset(); o.reset()
    t = 0
    for n in (80, 41, 100, 79, 200, 300):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)
        gm, om = g.mcts_state(), o.mcts_state()
        for name in gm.dtype.names:
            assert np.array_equal(gm[name], om[name]), (t, name)
        gr, orr = g.rewards(), o.rewards()
        assert np.array_equal(gr[0].view(np.uint32), orr[0].view(np.uint32)), t
        assert np.array_equal(gr[1].view(np.uint32), orr[1].view(np.uint32)), t


def test_training_mode_8_agents():
    import hierarchicalkarting_amd as hk
    A = 8
    b = hk.make_config(20, A, jitter_seed=0, env_mode=_lib.HK_MODE_TRAINING, rewards=1, training_agents=[1] * A, laps=1,
                       max_episode_steps=600)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    t = 0
    for n in (50, 150, 300, 300, 400):
        g.step(n); o.step(n); t += n
        _cmp(g, o, t)


def test_eight_agent_trajectory_hash_matches_the_committed_pin():
    """the kernels alone against the committed oracle pin (no oracle run on the GPU box)"""
    import hashlib, json, os
    import hierarchicalkarting_amd as hk
    from test_env_gpu import _pinned_bytes
    g = hk.RacingEnv(hk.make_config(2, 8, jitter_seed=0, laps=1))
    g.reset(experiment_num=0)
    hashes = []
    for _ in range(4):
        g.step(512)
        hashes.append(hashlib.sha256(_pinned_bytes(g.agent_state())).hexdigest())
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oval_8agent_2048_hash.json")))["sha256"]
    assert hashes == want
