"""Random call patterns against the oracle (round 6).  The host scheduler picks, per call, between lazy and fixed rounds, one batch and two halves, folded
arming, the optimistic plan, in-wave / queued / dense solves by the games meter, and regroups that pack or spread the envs that hold games — decisions that
depend on what the calls before left behind (meter words, beliefs, the order of the lane groups).  Whatever it picks, a getter must see the oracle's state
bit for bit.  Seeded random sequences of hk_step sizes (one tick to several hundred), looks at the state at random places, partial resets and full resets,
on a batch large enough to split (8 192 + 96 envs), with one-lap races and a short episode limit so that finishes, time-outs and auto-resets fall inside
the sequences."""
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

def cmp(g, o, tag):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (tag, name, np.argwhere(x != y)[:3].tolist())
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "episodes_done", "experiment_num", "status"):
        assert np.array_equal(ge[name], oe[name]), (tag, name)

seed = %(seed)d
rng = np.random.default_rng(seed)
E = 8192 + 96
b = hk.make_config(E, 4, jitter_seed=1000 + seed, laps=1, max_episode_steps=int(rng.choice([700, 1500, 4000])))
g = hk.RacingEnv(b); o = O.OracleEnv(b)
g.reset(); o.reset()
sizes = [1, 1, 1, 2, 3, 4, 5, 7, 8, 9, 16, 20, 20, 33, 63, 64, 65, 100, 130, 260, 520]
t = 0; k = 0; schedules = set()
while t < 2600:
    n = int(rng.choice(sizes))
    if rng.random() < 0.15:                       # a run of equal short calls (a host stepping tick by tick, the driver's window shape)
        for _ in range(int(rng.integers(5, 40))):
            g.step(n); o.step(n); t += n
    else:
        g.step(n); o.step(n); t += n
    s = g.schedule_info()
    schedules.add((s["rounds"], s["streams"], s["games_meter"], s["optimistic_plan"], s["armed_in_first_launch"]))
    r = rng.random()
    if r < 0.25:
        cmp(g, o, (seed, t, n))
    elif r < 0.29:
        ids = sorted(set(int(x) for x in rng.integers(0, E, size=int(rng.integers(1, 40)))))
        ex = int(rng.integers(0, 8))
        g.reset(ids, ex); o.reset(ids, ex)
    elif r < 0.31:
        g.reset(); o.reset()
    k += 1
g.synchronize()
cmp(g, o, (seed, t, "end"))
assert len(schedules) >= 2, schedules
print("fuzz ok", seed, t, k, sorted(schedules))
"""


@pytest.mark.parametrize("seed", [1, 3])
def test_random_call_patterns_match_the_oracle(seed):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HK_") or k in ("HK_LIB_PATH",)}
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "seed": seed}], env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
