"""The HIP environment (K_A begin / K_B solve / K_C move kernels through the C ABI) vs the CPU oracle on the same
seeded inputs: every field of every agent record bit-identical, tick by tick."""
import numpy as np
import pytest
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _cmp(gs, os_, ge, oe, tick):
    for name in gs.dtype.names:
        if not np.array_equal(gs[name], os_[name]):
            bad = np.argwhere(gs[name] != os_[name])
            e, a = bad[0][0], bad[0][1]
            raise AssertionError("tick %d field %s env %d agent %d: gpu %r oracle %r (%d mismatches)" % (
                tick, name, e, a, gs[name][e, a], os_[name][e, a], len(bad)))
    for name in ("episode_steps", "inactive_mask", "experiment_num", "episodes_done", "status", "initial_started"):
        assert np.array_equal(ge[name], oe[name]), (tick, name, ge[name][:8], oe[name][:8])


def _run(A, E, ticks, every, **kw):
    import hierarchicalkarting_amd as hk
    b = hk.make_config(E, A, **kw)
    g = hk.RacingEnv(b)
    o = O.OracleEnv(b)
    g.reset(); o.reset()
    _cmp(g.agent_state(), o.agent_state(), g.env_state(), o.env_state(), 0)
    t = 0
    while t < ticks:
        g.step(every); o.step(every); t += every
        _cmp(g.agent_state(), o.agent_state(), g.env_state(), o.env_state(), t)
    return g, o


def test_two_agent_tick_by_tick():
    _run(2, 4, 400, 1, jitter_seed=0x5EED0000)


def test_four_agent_tick_by_tick():
    _run(4, 8, 300, 1, jitter_seed=0x5EED0000)


def test_four_agent_long_many_envs():
    g, o = _run(4, 256, 1500, 250, jitter_seed=0x5EED0000)
    assert np.array_equal(g.observations(), o.observations())


def test_full_episode_with_auto_reset_two_agents():
    """4 096 ticks: a whole 2-agent race (~3.9 k ticks), the finish, the dead tick, the auto-reset and the next start."""
    g, o = _run(2, 16, 4096, 512, jitter_seed=0x5EED0000)
    gr, orr = g.episode_results(), o.episode_results()
    for name in gr.dtype.names:
        assert np.array_equal(gr[name], orr[name]), name
    assert (gr["episode"] >= 0).any()


def _pinned_bytes(st):
    """the 440 bytes per record the pin was taken over (px .. plan_vel); fields appended to hk_agent_state later
    (reward accumulators) stay out of it, so the pin keeps certifying the same trajectory"""
    raw = np.ascontiguousarray(st).view(np.uint8).reshape(st.shape + (st.dtype.itemsize,))
    return np.ascontiguousarray(np.concatenate([raw[..., :440], raw[..., 448:460]], axis=-1)).tobytes()    # ... and the engine's wheel state (ABI 5)


def test_trajectory_hash_matches_the_committed_pin():
    """BASELINE.md parity gate: 4 096-tick 2-agent Fixed-vs-Fixed Oval trajectory hash equal to the CPU oracle's."""
    import hashlib, json, os
    import hierarchicalkarting_amd as hk
    g = hk.RacingEnv(hk.make_config(1, 2, jitter_seed=0))
    g.reset(experiment_num=0)
    hashes = []
    for _ in range(8):
        g.step(512)
        hashes.append(hashlib.sha256(_pinned_bytes(g.agent_state())).hexdigest())
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oval_2agent_4096_hash.json")))["sha256"]
    assert hashes == want


def test_lq_debug_taps_and_decoded_controls(monkeypatch):
    """single-step a4 check: players, branch ids, targets, weights and u0 of every ego's game equal the oracle's"""
    monkeypatch.setenv("HK_LQ_DEBUG", "1")
    import hierarchicalkarting_amd as hk
    b = hk.make_config(16, 4, jitter_seed=0x5EED0000)
    g = hk.RacingEnv(b)
    o = O.OracleEnv(b)
    g.reset(); o.reset()
    seen = set()
    for _ in range(60):
        g.step(8); o.step(8)
        for env in (0, 5, 15):
            for ego in range(4):
                dg, do = g.lq_debug(env, ego), o.lq_debug(env, ego)
                assert dg.n_players == do.n_players
                for i in range(dg.n_players):
                    assert dg.player_agent[i] == do.player_agent[i] and dg.branch[i] == do.branch[i]
                    assert list(dg.initial[i]) == list(do.initial[i]) and list(dg.target[i]) == list(do.target[i])
                    assert list(dg.target_w[i]) == list(do.target_w[i]) and dg.control_w[i] == do.control_w[i]
                    seen.add(dg.branch[i])
                assert list(dg.u0) == list(do.u0)
    assert len(seen) >= 4, seen          # the race exercises several branches of the heading heuristic


def test_rl_actions_and_partial_reset():
    """LowMode == RL agents take hk_set_actions (KA:440-478); hk_reset of a subset leaves the other envs untouched"""
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd import _lib
    b = hk.make_config(6, 2, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_LQR], jitter_seed=7)
    g = hk.RacingEnv(b)
    o = O.OracleEnv(b)
    g.reset(); o.reset()
    rng = np.random.default_rng(0)
    for k in range(12):
        steer = rng.uniform(-1, 1, (6, 2)).astype(np.float32)
        branch = rng.integers(0, 3, (6, 2)).astype(np.int32)
        g.set_actions(steer, branch); o.set_actions(steer, branch)
        g.step(20); o.step(20)
        if k == 6:
            g.reset([1, 4], experiment_num=3); o.reset([1, 4], experiment_num=3)
        _cmp(g.agent_state(), o.agent_state(), g.env_state(), o.env_state(), k)
    assert (g.env_state()["experiment_num"][[1, 4]] == 3).all()


def test_invalid_configs_are_refused():
    import hierarchicalkarting_amd as hk
    from hierarchicalkarting_amd import _lib
    with pytest.raises(_lib.HkError) as e:
        hk.RacingEnv(hk.make_config(2, 4, high_mode=_lib.HK_HIGH_MCTS, tree_search_depth=12))   # gameParams.treeSearchDepth <= 8
    assert e.value.code == _lib.HK_ERR_INVALID
    with pytest.raises(_lib.HkError) as e:
        hk.RacingEnv(hk.make_config(2, 4, low_mode=_lib.HK_LOW_MPC))          # dead code in the reference
    assert e.value.code == _lib.HK_ERR_UNSUPPORTED
    with pytest.raises(_lib.HkError) as e:
        hk.RacingEnv(hk.make_config(2, 4, wiring=([0, 0, 1, 1], [[1], [0], [3], []], [[2, 3], [2, 3], [0, 1], [0, 1]])))
    assert e.value.code == _lib.HK_ERR_INVALID


def test_complex_track_four_agents():
    """the 41-section Complex track (straights, large / medium / small curves, S-curves), 3 laps,
    MaxLaneChanges 4; 842 wall segments through the LDS-staged wall grid vs the oracle's brute force."""
    g, o = _run(4, 24, 1200, 150, track="complex", jitter_seed=0x5EED0000)
    assert np.array_equal(g.observations(), o.observations())
    _run(2, 8, 600, 1, track="complex", jitter_seed=11)


def test_restored_env_words_cannot_arm_ticks_or_resume_a_phase():
    """hk_set_env_state stores the library's progress words as 0 / hint only (arming ADDS to reserved[0]): a snapshot restored with garbage in
    them — a record filled by hand — runs exactly the ticks of the next hk_step, as the oracle does"""
    import hierarchicalkarting_amd as hk
    b = hk.make_config(64, 4, jitter_seed=0x5EED0000)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    g.step(137); o.step(137)
    ag, es = g.agent_state(), g.env_state()
    bad = es.copy()
    bad["reserved"][:, 0] = 7                 # "seven ticks left"
    bad["reserved"][:, 1] = 1 | 16            # "waiting for controls", hint bit
    g2 = hk.RacingEnv(b); g2.reset()
    g2.set_agent_state(ag); g2.set_env_state(bad)
    back = g2.env_state()
    assert (back["reserved"][:, 0] == 0).all() and (back["reserved"][:, 1] == 16).all()
    for n in (1, 3, 60):
        g2.step(n); o.step(n)
        _cmp(g2.agent_state(), o.agent_state(), g2.env_state(), o.env_state(), 137 + n)
