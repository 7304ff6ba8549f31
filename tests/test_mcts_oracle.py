"""MCTS high-level planner, CPU oracle: structural properties of the plans, the timing of request ->
visible plan -> low-level targets, the beliefs about other karts, and race outcomes against the reference's published
experiment logs (MCTS-LQR beats Fixed-LQR more often than not; lap times in the same band)."""
import numpy as np
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.config import make_config

MC, FX = _lib.HK_HIGH_MCTS, _lib.HK_HIGH_FIXED


def _env(E, A, high, **kw):
    kw.setdefault("tree_search_depth", [8 if h == MC else 5 for h in high])
    b = make_config(E, A, high_mode=high, **kw)
    o = O.OracleEnv(b)
    o.reset()
    return o, b


def test_initial_plan_structure_and_latency():
    o, b = _env(6, 2, [MC, FX], jitter_seed=3, mcts_iterations=48)
    m = o.mcts_state()
    assert (m["ready_step"][:, 0] == 75).all() and (m["ready_step"][:, 1] == -1).all()
    assert (m["searches"][:, 0] == 1).all() and (m["best"]["n_states"] == 0).all()
    p = m["pend"][:, 0]
    assert (p["n_players"] == 2).all() and (p["n_states"] == 8).all()          # both karts within sectionWindow, depth 8
    a0 = o.agent_state()
    start = a0["section_index"].max(axis=1)                                     # initialSection = furthest kart's section
    for e in range(6):
        assert list(p["section"][e]) == list(range(start[e] + 1, start[e] + 9))
        assert set(np.unique(p["lane"][e][:, :2])) <= {1, 2, 3, 4}
        assert set(np.unique(p["vel"][e][:, :2])) <= {8, 10, 12, 14, 15}        # max_velocity of the 5 buckets
    assert (a0["plan_lane"][:, 0] == 0).all()                                   # nothing visible during the latency
    o.step(74)
    assert (o.agent_state()["plan_lane"][:, 0] == 0).all()
    o.step(1)                                                                   # episode step 75: pend -> best -> plan
    m = o.mcts_state(); a = o.agent_state()
    assert (m["best"]["n_states"][:, 0] == 8).all() and (m["ready_step"][:, 0] == -1).all()
    for e in range(6):
        pl = m["best"][e, 0]
        me = list(pl["player_agent"][:2]).index(0)
        for q in range(8):
            sec = int(pl["section"][q])
            if sec > a["section_index"][e, 0] + (0 if a["section_index"][e, 0] == 0 else 1):
                assert a["plan_lane"][e, 0][sec % 24] == pl["lane"][q][me]
                assert a["plan_vel"][e, 0][sec % 24] == float(pl["vel"][q][me])
            # belief about the other kart, keyed by section % L
            assert m["belief_lane"][e, 0][1][sec % 24] == pl["lane"][q][1 - me]
    # the Fixed agent holds no beliefs
    assert (m["belief_lane"][:, 1] == 0).all()


def test_replans_every_100_ticks_and_section_times():
    o, b = _env(3, 2, [MC, MC], jitter_seed=9, mcts_iterations=32)
    o.step(100)
    m = o.mcts_state()
    assert (m["searches"] == 2).all() and (m["ready_step"] == 145).all()
    o.step(45)
    assert (o.mcts_state()["ready_step"] == -1).all()
    o.step(155)                                                                 # step 300
    m = o.mcts_state(); a = o.agent_state()
    assert (m["searches"] == 4).all()
    for e in range(3):
        for i in range(2):
            s = int(a["section_index"][e, i])
            assert 0 < m["sec_time"][e, i][s & 3] <= 300                        # sectionTimes[m_SectionIndex] = tick of entry


def test_draws_are_keyed_by_env_and_agent_not_by_batch_shape():
    o1, _ = _env(4, 2, [MC, FX], jitter_seed=3, mcts_iterations=40)
    o2, _ = _env(2, 2, [MC, FX], jitter_seed=3, mcts_iterations=40, env_id_base=2)
    p1, p2 = o1.mcts_state()["pend"], o2.mcts_state()["pend"]
    for name in p1.dtype.names:
        assert np.array_equal(p1[name][2:], p2[name]), name


def test_races_against_the_reference_logs():
    """reference ExperimentLogs: MCTS-LQR vs Fixed-LQR on the Oval, 4 laps, finish ~3.9-4.1 k ticks and the MCTS
    agent wins the majority"""
    E = 12
    o, b = _env(E, 2, [MC, FX], jitter_seed=0x5EED0000, auto_reset=0, mcts_iterations=96)
    for _ in range(45):
        o.step(100)
        if (o.env_state()["inactive_mask"] == 3).all():
            break
    a = o.agent_state()
    t = a["time_steps"]
    assert ((t > 3600) & (t < 4400)).all(), t
    assert (t[:, 0] < t[:, 1]).mean() >= 0.5
    assert (a["illegal_lane_changes"] <= 6).all()


def test_four_agents_two_teams_with_mcts():
    o, b = _env(4, 4, [MC, MC, FX, FX], jitter_seed=11, mcts_iterations=24)
    o.step(400)
    m = o.mcts_state()
    assert (m["searches"][:, :2] >= 4).all() and (m["searches"][:, 2:] == 0).all()
    assert (m["best"]["n_players"][:, :2] >= 1).all()
    assert np.isfinite(o.agent_state()["px"]).all()
