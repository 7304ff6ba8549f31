"""RL low-level policy on device (policy_stack_kernel + policy_mlp_kernel on the f32 MFMA, through the C ABI) vs the CPU
oracle: mu / logits bit-identical (the MFMA result is a k-ascending fmaf chain seeded with the bias, which is what the
oracle evaluates), sampled actions bit-identical (same Philox draws), and whole RL-driven episodes bit-identical."""
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.policy import Policy

pytestmark = pytest.mark.gpu


def _pair(E, A, low, **kw):
    import hierarchicalkarting_amd as hk
    b = hk.make_config(E, A, low_mode=low, **kw)
    g = hk.RacingEnv(b)
    o = O.OracleEnv(b)
    g.reset(); o.reset()
    return g, o


@pytest.mark.parametrize("A,stack,hidden,layers,rows,normalize", [
    (2, 4, 128, 3, 1000, True),      # HierarchicalAgent-NonLSTM-* shape (216 -> 128 x 3)
    (4, 4, 256, 3, 777, True),       # HierarchicalAgent-Team-*scaledown* shape (312 -> 256 x 3)
    (4, 8, 256, 3, 130, True),       # HierarchicalAgent-Team-all* shape (624 -> 256 x 3): layer 0 in two chunks
    (2, 4, 64, 1, 64, False),
    (2, 4, 192, 2, 1, True),
    (2, 4, 32, 4, 65, True),
    (4, 1, 256, 2, 100, True),       # 78 inputs: 9 whole groups of 8 (the float4 weight copy) + 6 through the per-k-step remainder
    (4, 7, 256, 2, 70, True),        # 546 inputs: three chunks of 184 / 184 / 178 (22 groups + 2)
    (2, 3, 96, 2, 50, False),        # 162 inputs, 3 column blocks on 8 waves
])
def test_actor_bit_exact(A, stack, hidden, layers, rows, normalize):
    g, o = _pair(2, A, [_lib.HK_LOW_RL] * A)
    in_dim = g.obs_dim * stack
    pol = Policy.random(in_dim, hidden, layers, stack=stack, seed=hidden + rows, normalize=normalize)
    gi = g.attach_policy(pol, [0], 2)
    oi = o.attach_policy(pol, [0], 2)
    r = np.random.default_rng(rows)
    obs = (r.standard_normal((rows, in_dim)) * 4).astype(np.float32)
    obs[0, :7] = [0.0, -0.0, 1e-30, -1e30, 5.0, -5.0, 1e30]
    gm, gl = g.policy_forward(gi, obs)
    om, ol = o.policy_forward(oi, obs)
    assert np.array_equal(gm.view(np.uint32), om.view(np.uint32)), np.abs(gm - om).max()
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), np.abs(gl - ol).max()
    assert np.isfinite(gm).all() and np.abs(gm).max() > 1e-3


def _cmp_agents(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        assert np.array_equal(gs[name], os_[name]), (t, name)
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "experiment_num", "episodes_done", "status", "initial_started"):
        assert np.array_equal(ge[name], oe[name]), (t, name)
    ga, oa = g.get_actions(), o.get_actions()
    assert np.array_equal(ga[0].view(np.uint32), oa[0].view(np.uint32)), (t, "steer")
    assert np.array_equal(ga[1], oa[1]), (t, "branch")


def test_rl_vs_lq_episode_tick_by_tick():
    """agent 0 driven by a (stochastic) actor, agent 1 by the LQ planner; 2-tick decisions; start hold included"""
    g, o = _pair(6, 2, [_lib.HK_LOW_RL, _lib.HK_LOW_LQR], jitter_seed=21)
    pol = Policy.random(g.obs_dim * 4, 128, 3, seed=5)
    g.attach_policy(pol, [0], 2); o.attach_policy(pol, [0], 2)
    seen = set()
    for t in range(1, 241):
        g.step(1); o.step(1)
        _cmp_agents(g, o, t)
        s, br = g.get_actions()
        seen.update(np.unique(s[:, 0]).tolist())
        assert (s[:, 1] == 0).all()                      # nothing latches an action for the LQ agent
    assert len(seen) > 6                                 # the actor's (sampled) steering varied over the episode and the envs


def test_two_team_policies_with_timeout_resets():
    """2v2, every agent RL, one actor per team (as the reference's Team 1 / Team 2 models), stack 8 for team 2 is not
    possible on one observation size so both use 4; odd step sizes cross decision boundaries; max_episode_steps = 150
    forces time-outs -> auto-reset -> the observation stacks restart from zeros"""
    g, o = _pair(40, 4, [_lib.HK_LOW_RL] * 4, jitter_seed=4, max_episode_steps=150)
    p1 = Policy.random(g.obs_dim * 4, 256, 3, seed=1)
    p2 = Policy.random(g.obs_dim * 4, 128, 2, seed=2, deterministic=True)
    for e in (g, o):
        assert e.attach_policy(p1, [0, 1], 2) == 0
        assert e.attach_policy(p2, [2, 3], 2) == 1
    t = 0
    for n in (1, 3, 7, 2, 50, 101, 33, 64, 150, 75):
        g.step(n); o.step(n); t += n
        _cmp_agents(g, o, t)
    assert (g.env_state()["episodes_done"] >= 2).all()
    assert np.array_equal(g.observations(), o.observations())


def test_decision_period_and_explicit_reset():
    g, o = _pair(5, 2, [_lib.HK_LOW_RL] * 2, jitter_seed=8)
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=12)
    g.attach_policy(pol, [0, 1], 3); o.attach_policy(pol, [0, 1], 3)
    for t in range(1, 20):
        g.step(1); o.step(1)
        _cmp_agents(g, o, t)
    g.reset([1, 3], 1); o.reset([1, 3], 1)
    for t in range(20, 40):
        g.step(1); o.step(1)
        _cmp_agents(g, o, t)


def test_attach_rejects_bad_requests():
    import hierarchicalkarting_amd as hk
    g, o = _pair(2, 2, [_lib.HK_LOW_RL, _lib.HK_LOW_LQR])
    good = Policy.random(g.obs_dim * 4, 64, 1)
    with pytest.raises(hk.HkError) as e:
        g.attach_policy(good, [1])                       # not an RL agent
    assert e.value.code == _lib.HK_ERR_INVALID
    with pytest.raises(hk.HkError):
        g.attach_policy(Policy.random(g.obs_dim * 4 + 2, 64, 1), [0])   # trained on another observation size
    with pytest.raises(hk.HkError):
        g.attach_policy(Policy.random(g.obs_dim * 4, 48, 1), [0])       # hidden not a multiple of 32
    assert g.attach_policy(good, [0]) == 0
    with pytest.raises(hk.HkError):
        g.attach_policy(good, [0])                       # slot already driven
