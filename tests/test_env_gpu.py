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
