"""In-wave solves (round 6: env_b1_kernel solves the multi-player games its own waves assembled, hk_lq_spread.h lqs_inwave) against the oracle, every field
bit for bit, on fields built to hold games of EVERY size the quad handles meet: karts packed behind the first Trigger in clusters of two, three and four
(two 2-player games side by side in a wave's slice of the staging area, a 3-player game alone in it, 4-player games through the block's first wave behind
the block barrier), in the three places the games can be solved (HK_INWAVE=1: in-wave in every round; HK_INWAVE=0: queues + the spread solver's launch;
HK_INWAVE=0 HK_LQN=pair: queues + the pair / matrix-core kernel).  The switches are read in hk_create, so each runs in a child process."""
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

def cmp(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name, np.argwhere(x != y)[:3].tolist())

E = 8192 + 192 if os.environ.get("HK_SPLIT") else 200
b = hk.make_config(E, 4, jitter_seed=21, laps=1, max_episode_steps=1200)
os.environ["HK_LQ_DEBUG"] = "1"
g = hk.RacingEnv(b); o = O.OracleEnv(b)
del os.environ["HK_LQ_DEBUG"]
g.reset(); o.reset()
st = o.agent_state().copy()
s0 = b.track["sections"][0]
# env mod 4: 0 -> all four karts in one cluster (4-player games), 1 -> three + one far ahead, 2 -> two pairs 30 m apart, 3 -> the grid as reset left it
for env in range(E):
    kind = env %% 4
    if kind == 3:
        continue
    for j in range(4):
        lane = j %% 4 + 1
        far = (kind == 1 and j == 3) or (kind == 2 and j >= 2)
        st["px"][env, j] = s0["Lane%%d" %% lane]["x"]
        st["pz"][env, j] = s0["Lane%%d" %% lane]["z"] + 2.0 + (30.0 if far else 0.0) + 0.01 * (env %% 7)
        st["lane"][env, j] = lane
g.set_agent_state(st); o.set_agent_state(st)
g.prof_enable(True); g.prof_reset()
seen = set()
t = 0
for n in (76, 4, 4, 8, 1, 3, 20, 64, 120, 300):
    g.step(n); o.step(n); t += n
    cmp(g, o, t)
    for env in range(0, 8):
        for ego in range(4):
            gd, od = g.lq_debug(env, ego), o.lq_debug(env, ego)
            seen.add(int(od.n_players))
            assert gd.n_players == od.n_players and gd.u0[0] == od.u0[0] and gd.u0[1] == od.u0[1], (t, env, ego, gd.n_players, od.n_players, gd.u0[0], od.u0[0])
games = g.prof_games()
assert {2, 3, 4} <= seen, seen
assert games[2] > 0 and games[3] > 0 and games[4] > 0, games
print("inwave ok", games)
"""

MODES = {"inwave": {"HK_INWAVE": "1"}, "inwave_split": {"HK_INWAVE": "1", "HK_SPLIT": "1"}, "inwave_tab_global": {"HK_INWAVE": "1", "HK_TAB_GLOBAL": "1"},
         "default": {}, "queues": {"HK_INWAVE": "0"}, "queues_pair": {"HK_INWAVE": "0", "HK_LQN": "pair"}}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_games_of_every_size_wherever_they_are_solved(mode):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HK_") or k in ("HK_LIB_PATH",)}
    env.update(MODES[mode])
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "inwave ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
