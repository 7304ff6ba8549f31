"""Training mode (REC.ResetGame :520-668 random scatter, HKA.planRandomly :109-143), CPU oracle: structural properties."""
import numpy as np
import oracle_lib as O
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.config import make_config


def test_random_scatter_properties():
    b = make_config(400, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 1, 1], laps=2, jitter_seed=0)
    o = O.OracleEnv(b); o.reset()
    a = o.agent_state()
    sec, lane = a["section_index"], a["lane"]
    goal = 2 * 24 + 1
    assert sec.min() >= 0 and sec.max() < goal and set(np.unique(lane)) == {1, 2, 3, 4}
    assert (a["init_checkpoint_index"] == sec).all()
    spread = sec.max(axis=1) - sec.min(axis=1)
    h2h = spread <= 3                                   # head to head: everyone within [first - 1, first + 1]
    assert 0.45 < h2h.mean() < 0.85                     # Random.Range(0, 9) >= 3 -> 2/3 (plus scattered envs that happen to be close)
    assert (spread > 6).any()                           # ... and genuinely scattered ones exist
    # no two karts on the same lane marker
    for e in range(400):
        keys = {(int(s) % 24, int(l)) for s, l in zip(sec[e], lane[e])}
        assert len(keys) == 4
    # tire wear drawn in [0, 1): acc_ang_v = -TireWearRate * log(1 - 0.75 * twp) in [0, ~13 863)
    assert a["acc_ang_v"].min() >= 0 and a["acc_ang_v"].max() < 13900 and a["acc_ang_v"].std() > 1000
    # karts can move at once (no 1.5 s hold in Training mode) and carry random plans of 5 sections
    o.step(1)
    assert ((o.agent_state()["flags"] & _lib.HK_F_CAN_MOVE) != 0).all()
    pl = a["plan_lane"]
    assert ((pl != 0).sum(axis=2) == 5).all()
    # planRandomly is biased to the optimal side: index 0 (|N(0,1)| rounds to 0 in 38 % of the draws) is the most frequent
    assert np.isin(pl[pl != 0], [1, 2, 3, 4]).all()


def test_training_episodes_run_and_reset_differently():
    b = make_config(16, 2, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1], laps=1, max_episode_steps=400, rewards=1, jitter_seed=0)
    o = O.OracleEnv(b); o.reset()
    first = o.agent_state()["section_index"].copy()
    o.step(450)
    es = o.env_state()
    assert (es["episodes_done"] >= 1).all()
    second = o.agent_state()
    assert np.isfinite(second["px"]).all()
    o.step(400)
    third = o.agent_state()["init_checkpoint_index"]
    assert (third != first).any()                        # a new episode draws a new layout


def test_mcts_agents_in_training_mode_plan_randomly():
    b = make_config(8, 2, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 0], high_mode=[_lib.HK_HIGH_MCTS, _lib.HK_HIGH_MCTS],
                    tree_search_depth=8, mcts_iterations=8, jitter_seed=0)
    o = O.OracleEnv(b); o.reset()
    m = o.mcts_state()
    assert (m["searches"][:, 0] == 0).all() and (m["searches"][:, 1] == 1).all()
    a = o.agent_state()
    assert ((a["plan_lane"][:, 0] != 0).sum(axis=1) == 8).all()           # planRandomly filled depth 8 at once
    v = a["plan_vel"][:, 0][a["plan_lane"][:, 0] != 0]
    assert (v <= 15.0).all() and (v >= 7.0).all() and v.std() > 0.1     # max speed - |N(0, 1.5)|, clipped at 8
