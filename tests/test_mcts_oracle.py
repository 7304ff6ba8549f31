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
            assert 0 < m["sec_time"][e, i][s & 7] <= 300                        # sectionTimes[m_SectionIndex] = tick of entry


def test_root_reuse_follows_the_reference_rules():
    """HKA:66-67,175,265-283,660-669: a replan searches the tree of the previous plan again (CyclesRootProcessed < 3) unless the
    kart entered a section after that plan was finished; a search that is in flight when the root is dropped re-installs it."""
    o, b = _env(8, 2, [MC, MC], jitter_seed=9, mcts_iterations=24)
    m = o.mcts_state()
    assert (m["pend_kind"] == 1).all() and (m["root_phases"] == 1).all() and (m["root_live"] == 0).all()
    o.step(75)                                                                  # the first plan is finished: currentRoot set
    m = o.mcts_state()
    assert (m["root_live"] == 1).all() and (m["root_cycles"] == 1).all() and (m["pend_kind"] == 0).all()
    first = m["best"].copy()
    sec75 = o.agent_state()["section_index"].copy()
    o.step(25)                                                                  # tick 100: the replan
    m = o.mcts_state(); a = o.agent_state()
    moved = a["section_index"] != sec75                                         # entered a section since tick 75 -> root dropped
    assert (~moved).any(), "the test needs karts that are still in their start section at tick 100"
    assert (m["pend_kind"][~moved] == 2).all() and (m["root_phases"][~moved] == 2).all()
    assert (m["pend_kind"][moved] == 1).all() and (m["root_phases"][moved] == 1).all()
    # a re-searched root keeps its position: the plan still starts right after the ORIGINAL initial section
    for e, i in zip(*np.nonzero(~moved)):
        assert m["pend"]["section"][e, i][0] == first["section"][e, i][0]
        assert list(m["pend"]["player_agent"][e, i]) == list(first["player_agent"][e, i])
    live100 = m["root_live"].copy()
    assert (live100[~moved] == 1).all() and (live100[moved] == 0).all()
    o.step(45)                                                                  # tick 145: the second search is finished
    m2 = o.mcts_state(); a2 = o.agent_state()
    assert (m2["root_live"] == 1).all() and (m2["pend_kind"] == 0).all()        # whatever happened in between, the thread wrote currentRoot
    passed = a2["section_index"] != a["section_index"]
    want = np.where(~moved, np.where(passed, 1, 2), 1)                          # REUSE: cycles (reset to 0 by a pass) + 1; NEW: 1
    assert np.array_equal(m2["root_cycles"], want)
    # a moving kart passes a section between two replans: from tick 200 on every replan builds a new tree
    o.step(255)                                                                 # tick 400
    m3 = o.mcts_state()
    assert (m3["root_phases"] == 1).all() and (m3["searches"] == 5).all()


def test_stuck_kart_stops_replanning_after_three_searches_of_one_tree():
    """a kart that never enters a section: tree searched at reset, tick 100, tick 200 (CyclesRootProcessed 1, 2, 3), then no
    search at all (HKA:265 is false and :175 is false)"""
    o, b = _env(2, 2, [MC, FX], mcts_iterations=16)
    st = o.agent_state()
    for t in range(20):
        o.step(25)
        a = o.agent_state()
        a[:, 0] = st[:, 0]                                                      # kart 0 back on its grid slot, still held (m_CanMove false)
        o.set_agent_state(a)
    m = o.mcts_state()
    assert (o.agent_state()["section_index"][:, 0] == st["section_index"][:, 0]).all()
    assert (m["searches"][:, 0] == 3).all() and (m["root_cycles"][:, 0] == 3).all() and (m["root_phases"][:, 0] == 3).all()


def test_reset_backfills_section_times_and_reads_stale_steer():
    """REC:679-702: a kart that starts ahead of the rearmost one gets drawn (negative, descending) section times for the sections
    behind it; HKA:221-224 turns them into the time offsets of the first plan.  REC:705-710: the first plan of agent i sees the
    tire age of agents j > i as the previous episode left it."""
    o, b = _env(5, 4, [MC, MC, MC, MC], jitter_seed=4, mcts_iterations=16, mcts_seed=77)
    a = o.agent_state(); m = o.mcts_state()
    for e in range(5):
        back = a["section_index"][e].min()
        for i in range(4):
            s = int(a["section_index"][e, i])
            assert m["sec_time"][e, i][s & 7] == 0
            prev = -b.cfg.max_episode_steps
            for tp in range(back, s):
                v = int(m["sec_time"][e, i][tp & 7])
                assert prev <= v < 0
                prev = v
    # a second handle with another seed draws other times, and its first plans differ somewhere
    o2, _ = _env(5, 4, [MC, MC, MC, MC], jitter_seed=4, mcts_iterations=16, mcts_seed=78)
    assert not np.array_equal(o2.mcts_state()["sec_time"], m["sec_time"])
    # stale steer: two handles whose karts carry different m_FinalStats.Steer when ResetGame runs.  Agent 1 plans after its own and
    # agent 0's prepareForReuse (fresh values for both): same plans.  Agent 0 plans before agent 1's: it reads the stale value.
    def after_reset(steer1):
        oo, _ = _env(16, 2, [MC, MC], jitter_seed=4, mcts_iterations=32)
        st = oo.agent_state()
        st["final_steer"][:, 1] = steer1
        oo.set_agent_state(st)
        oo.reset()
        return oo.mcts_state()["pend"]
    pa, pb = after_reset(4.0), after_reset(1.0)                                 # tire age 0 vs 10 000 as agent 0 sees agent 1
    for name in pa.dtype.names:
        assert np.array_equal(pa[name][:, 1], pb[name][:, 1]), name
    assert any(not np.array_equal(pa[name][:, 0], pb[name][:, 0]) for name in ("lane", "vel"))


def test_draws_are_keyed_by_env_and_agent_not_by_batch_shape():
    o1, _ = _env(4, 2, [MC, FX], jitter_seed=3, mcts_iterations=40)
    o2, _ = _env(2, 2, [MC, FX], jitter_seed=3, mcts_iterations=40, env_id_base=2)
    p1, p2 = o1.mcts_state()["pend"], o2.mcts_state()["pend"]
    for name in p1.dtype.names:
        assert np.array_equal(p1[name][2:], p2[name]), name


def test_races_against_the_reference_logs():
    """reference ExperimentLogs: MCTS-LQR vs Fixed-LQR on the Oval, 4 laps, finish ~3.9-4.1 k ticks and the MCTS
    agent wins its share (24 of 50 in the reference's log MCTS_LQR_vs_Fixed_LQR_Oval2)"""
    E = 12
    o, b = _env(E, 2, [MC, FX], jitter_seed=0x5EED0000, auto_reset=0, mcts_iterations=96)
    for _ in range(45):
        o.step(100)
        if (o.env_state()["inactive_mask"] == 3).all():
            break
    a = o.agent_state()
    t = a["time_steps"]
    assert ((t > 3600) & (t < 4400)).all(), t
    assert 0.25 <= (t[:, 0] < t[:, 1]).mean() <= 0.9
    assert (a["illegal_lane_changes"] <= 6).all()


def test_four_agents_two_teams_with_mcts():
    o, b = _env(4, 4, [MC, MC, FX, FX], jitter_seed=11, mcts_iterations=24)
    o.step(400)
    m = o.mcts_state()
    assert (m["searches"][:, :2] >= 4).all() and (m["searches"][:, 2:] == 0).all()
    assert (m["best"]["n_players"][:, :2] >= 1).all()
    assert np.isfinite(o.agent_state()["px"]).all()
