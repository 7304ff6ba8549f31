"""Parity of every instantiation of the tick kernel under every scheduling mode (a kernel whose results once depended
on its loop form deserves a test per instantiated <HAS_MCTS, HAS_RW, HAS_TRAIN, TAB_LDS> x {eager assembly on / off, planner pause on / off,
split batch, fused / fission}) and of the 8-lane groups' instantiations (hk::g8).  The switches are read once per process (hk_create), so every combination runs in a child process; each child steps a
small batch through resets, short and long calls against the CPU oracle, every field of every agent record bit for bit."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
MC, FX, LQ = _lib.HK_HIGH_MCTS, _lib.HK_HIGH_FIXED, _lib.HK_LOW_LQR
kind = %(kind)r
kw = dict(jitter_seed=0x5EED0000, laps=1, max_episode_steps=260)
E = 8192 + 64 if os.environ.get("HK_SPLIT") == "1" else 160
if kind == "plain":          cfg = hk.make_config(E, 4, **kw)                                              # <false, false, false>
elif kind == "rewards":      cfg = hk.make_config(E, 4, rewards=1, **kw)                                   # <false, true, false>
elif kind == "planner":      cfg = hk.make_config(E, 4, high_mode=[MC, MC, FX, FX], tree_search_depth=[8, 8, 5, 5], mcts_iterations=12, **kw)          # <true, false, false>
elif kind == "planner_rw":   cfg = hk.make_config(E, 4, high_mode=[MC, FX, MC, FX], tree_search_depth=[8, 5, 8, 5], mcts_iterations=12, rewards=1, **kw)  # <true, true, false>
elif kind == "training":     cfg = hk.make_config(E, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], rewards=1, jitter_seed=0, laps=1, max_episode_steps=260)   # <true, true, true>
# round 4: handles with attached actors — with and without LQ agents beside them (decision chunks on the fission schedule / the tick kernel alone)
elif kind == "actor_lq":     cfg = hk.make_config(E, 4, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_RL, LQ, LQ], **kw)
elif kind == "actor_only":   cfg = hk.make_config(E, 4, low_mode=[_lib.HK_LOW_RL] * 4, **kw)
elif kind == "planner_actor_lq": cfg = hk.make_config(E, 4, high_mode=[MC, MC, FX, FX], low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_RL, LQ, LQ], tree_search_depth=[8, 8, 5, 5], mcts_iterations=12, **kw)
# the 8-lane groups (hk::g8, BASELINE configs[4]): plain LQNG, planner + on-device actor for one team, Training mode
E8 = 40
RL = _lib.HK_LOW_RL
if kind == "g8_plain":       cfg = hk.make_config(E8, 8, **kw)
elif kind == "g8_planner_actor": cfg = hk.make_config(E8, 8, high_mode=[MC] * 8, low_mode=[RL] * 4 + [LQ] * 4, tree_search_depth=8, mcts_iterations=8, **kw)
elif kind == "g8_training":  cfg = hk.make_config(E8, 8, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1] * 4 + [0] * 4, rewards=1, jitter_seed=0, laps=1, max_episode_steps=260)
g = hk.RacingEnv(cfg); o = O.OracleEnv(cfg)
if kind in ("actor_lq", "actor_only", "planner_actor_lq"):
    from hierarchicalkarting_amd.policy import Policy
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=78)
    slots = [0, 1] if kind != "actor_only" else [0, 1, 2, 3]
    g.attach_policy(pol, slots, 2); o.attach_policy(pol, slots, 2)
if kind == "g8_planner_actor":
    from hierarchicalkarting_amd.policy import Policy
    pol = Policy.random(g.obs_dim * 4, 64, 2, seed=77)
    g.attach_policy(pol, [0, 1, 2, 3], 2); o.attach_policy(pol, [0, 1, 2, 3], 2)
g.reset(); o.reset()
t = 0
for n in ((90, 70, 1, 2, 3, 20, 7, 107, 300) if not kind.startswith("g8") else (90, 40, 1, 2, 3, 20, 7, 127)):   # start hold, race start, short calls, long calls (lazy / pause), a time-out reset inside
    g.step(n); o.step(n); t += n
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (kind, t, name, np.argwhere(x != y)[:3].tolist())
assert (g.env_state()["episodes_done"] >= 1).all() or kind.startswith("g8") or "actor" in kind
print("matrix ok", kind)
"""

MODES = {"default": {}, "tab_global": {"HK_TAB_GLOBAL": "1"}, "fixed_rounds": {"HK_FIXED_ROUNDS": "1"}}
CASES = [(k, m) for k in ("plain", "rewards", "planner", "planner_rw", "training") for m in MODES]
CASES += [("plain", "split"), ("rewards", "split"), ("planner", "no_pause"), ("planner_rw", "no_pause"), ("training", "no_pause")]
# the fused tick kernel of plain handles (the default since round 4 is the tick kernel without phase B1 + env_b1_kernel: hk_env_run.h FISSION)
CASES += [("plain", "fused"), ("plain", "fused_split"), ("plain", "fused_tab_global")]
# the 8-lane groups under the scheduling modes
CASES += [(k, m) for k in ("g8_plain", "g8_planner_actor", "g8_training") for m in ("default", "fixed_rounds", "tab_global")]
# the fission schedule of planner / actor handles against its fused alternatives (round 4)
CASES += [(k, m) for k in ("actor_lq", "actor_only", "planner_actor_lq") for m in ("default", "fused", "tab_global")]
CASES += [("planner", "fused")]
# round 5: the planner's long calls with the searches beside the ticks (the default) against the schedule that stops an env at its request
CASES += [("planner", "no_overlap"), ("planner_rw", "no_overlap")]
# reward-shaped / Training handles: the fission schedule is their default since round 5; their fused kernel stays reachable (HK_FISSION=0: every handle fused)
CASES += [("rewards", "fused"), ("planner_rw", "fused"), ("training", "fused")]
MODES.update({"no_overlap": {"HK_MCTS_NO_OVERLAP": "1"}})
# the searches beside the ticks in 8-wave workgroups on half the CUs (the default is 4 waves on every CU wherever a tick block fits beside one, with phase B1 on
# its global-table instantiation and lqn_round_small_kernel for those rounds): both forms of the side launch stay reachable
CASES += [("planner", "side8"), ("planner_rw", "side8")]
MODES.update({"side8": {"HK_MCTS_SIDE_WAVES": "8"}})
# round 6: where the multi-player games of the fission schedule are solved — by the B1 waves that assembled them (in every round, the race start too; the
# default does so only once the field has spread), through the queues by the spread solver's launch, through the queues by the pair / matrix-core kernel
CASES += [(k, m) for k in ("plain", "rewards", "planner", "planner_rw", "training", "actor_lq", "planner_actor_lq") for m in ("inwave_all", "queues", "queues_pair")]
CASES += [("plain", "inwave_all_split"), ("plain", "inwave_all_tab_global")]
MODES.update({"inwave_all": {"HK_INWAVE": "1"}, "queues": {"HK_INWAVE": "0"}, "queues_pair": {"HK_INWAVE": "0", "HK_LQN": "pair"},
              "inwave_all_split": {"HK_INWAVE": "1", "HK_SPLIT": "1"}, "inwave_all_tab_global": {"HK_INWAVE": "1", "HK_TAB_GLOBAL": "1"}})
MODES.update({"split": {"HK_SPLIT": "1"}, "no_pause": {"HK_MCTS_NO_PAUSE": "1"}, "fused": {"HK_FISSION": "0"}, "fused_split": {"HK_FISSION": "0", "HK_SPLIT": "1"},
              "fused_tab_global": {"HK_FISSION": "0", "HK_TAB_GLOBAL": "1"}})


@pytest.mark.parametrize("kind,mode", CASES)
def test_every_instantiation_under_every_mode(kind, mode):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HK_") or k in ("HK_LIB_PATH",)}
    env.update(MODES[mode])
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "kind": kind}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "matrix ok " + kind in r.stdout
