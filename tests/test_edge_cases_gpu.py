"""Edge cases at the boundary: empty / minimal inputs and maximum table sizes."""
import copy
import numpy as np
import pytest
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.config import load_track

pytestmark = pytest.mark.gpu


def _same(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name)


def test_zero_ticks_single_env_and_empty_batches():
    import hierarchicalkarting_amd as hk
    g = hk.RacingEnv(hk.make_config(1, 1, wiring=([0], [[]], [[]]), jitter_seed=1))
    g.reset()
    before = g.agent_state().tobytes()
    g.step(0)
    assert g.agent_state().tobytes() == before
    g.reset([], 0)                                     # empty id list: nothing happens
    assert g.agent_state().tobytes() == before
    u = hk.solve_feedback_lqr_batch(np.zeros((0, 2, 4, 4)), np.zeros((0, 2, 4, 2)), np.zeros((0, 2, 8, 8)), np.zeros((0, 2, 8)),
                                    np.zeros((0, 2, 2, 2)), np.zeros((0, 8)), 3)
    assert u.shape == (0, 2)


def test_track_without_walls():
    """open field: every ray reports its maximum distance, no wall contacts; still identical to the oracle"""
    import hierarchicalkarting_amd as hk
    tr = copy.deepcopy(load_track("oval"))
    tr["walls"] = []
    b = hk.make_config(5, 2, track=tr, jitter_seed=2)
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    for t in (100, 200, 300):
        g.step(100); o.step(100)
        _same(g, o, t)
    obs = g.observations()
    assert np.array_equal(obs, o.observations())


def test_track_tables_too_large_for_lds_are_read_from_global_memory():
    """the tick kernel keeps the track tables in LDS when they fit 48 KB (a compile-time choice: env_run_kernel<..., TAB_LDS>);
    a track with finely tessellated walls takes the other instantiation — same results as the oracle on the same walls"""
    import hierarchicalkarting_amd as hk
    tr = copy.deepcopy(load_track("oval"))
    k = 12
    for w in tr["walls"]:
        pts, out = w["points"], []
        for (x0, z0), (x1, z1) in zip(pts[:-1], pts[1:]):
            for j in range(k):
                out.append([x0 + (x1 - x0) * j / k, z0 + (z1 - z0) * j / k])
        out.append(list(pts[-1]))
        w["points"] = out
    b = hk.make_config(6, 4, track=tr, jitter_seed=5)
    assert b.cfg.num_walls * 16 > 48 * 1024                                    # the wall segments alone exceed the LDS budget
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    for t in (100, 200, 300, 400):
        g.step(100); o.step(100)
        _same(g, o, t)
    assert np.array_equal(g.observations(), o.observations())


def test_maximum_section_count_and_many_laps():
    """HK_MAX_SECTIONS = 64 sections (the Complex track's 41 repeated would exceed it: take the Oval's 24 x 2 + 16 = 64) and a
    long race (reward tables sized laps * L + 2)"""
    import hierarchicalkarting_amd as hk
    tr = copy.deepcopy(load_track("oval"))
    secs = tr["sections"]
    tr["sections"] = (secs + secs + secs)[:64]          # geometry repeats: karts lap the same oval, the index space is 64 long
    b = hk.make_config(4, 2, track=tr, jitter_seed=3, laps=9, rewards=1, max_episode_steps=900)
    assert b.cfg.num_sections == 64
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    for t in (300, 600, 950):
        g.step(t if t == 300 else (300 if t == 600 else 350)); o.step(t if t == 300 else (300 if t == 600 else 350))
        _same(g, o, t)
    with pytest.raises(hk.HkError):
        tr2 = copy.deepcopy(tr); tr2["sections"] = tr["sections"] + secs[:1]       # 65 sections
        hk.RacingEnv(hk.make_config(2, 2, track=tr2))


@pytest.mark.parametrize("A", [2, 4, 8])
def test_short_calls_across_resets_need_no_more_rounds_than_issued(A):
    """hk_step(1), (2), (3) ... issue a fixed number of rounds without looking at the device (env_rounds_for: solve ticks in the call + 1).
    Episodes of at most 120 ticks put time-out resets — whose reset tick is a solve tick of its own — inside many of the calls; a call that
    needed one round more would raise "an env did not complete its ticks" at the next getter, a wrong state would differ from the oracle."""
    import hierarchicalkarting_amd as hk
    b = hk.make_config(96, A, jitter_seed=0x5EED0000, laps=1, max_episode_steps=100 + 7 * (A % 3))
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    t = 0
    for n, reps in ((1, 130), (2, 60), (3, 45), (4, 30), (5, 25), (7, 18), (20, 7), (33, 4)):
        for _ in range(reps):
            g.step(n); o.step(n); t += n
        gs, os_ = g.agent_state(), o.agent_state()
        for name in gs.dtype.names:
            x, y = gs[name], os_[name]
            if x.dtype.kind == "f":
                x = x.view(np.uint32); y = y.view(np.uint32)
            assert np.array_equal(x, y), (A, t, n, name)
    assert (g.env_state()["episodes_done"] >= 5).all()
