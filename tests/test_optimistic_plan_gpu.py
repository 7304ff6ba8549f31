"""The optimistic round plan of fixed-round calls (hk_api.hip step_ticks, round 5): a plain handle whose envs are believed to stand on the same episode step
gets exactly the launches a field in lock-step needs; the completion guard verifies the belief and the next entry point that looks at the state finishes any
env the plan missed.  Whatever the belief, the state a getter sees must be the oracle's, bit for bit: with a right belief, with a belief that is wrong by
construction (HK_OPTIMISTIC_SKEW: every plan misses solve ticks), with one that becomes wrong (time-outs and finishes inside short calls), and without it."""
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
    ge, oe = g.env_state(), o.env_state()
    for name in ("episode_steps", "inactive_mask", "episodes_done", "experiment_num", "status"):
        assert np.array_equal(ge[name], oe[name]), (t, name)

def run(b, calls, look_every):
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    t = 0
    for k, n in enumerate(calls):
        g.step(n); o.step(n); t += n
        if (k + 1) %% look_every == 0:
            cmp(g, o, t)
    g.synchronize()
    cmp(g, o, t)
    return g, o

E = 8192 + 128 if os.environ.get("HK_SPLIT") == "1" else 192
# a host that steps tick by tick through the start hold and the race start, looking only now and then; then the driver's call shape; then odd sizes
g, o = run(hk.make_config(E, 4, jitter_seed=0x5EED0000, laps=1), [1] * 100 + [5, 20, 20, 3, 1, 2, 7, 20, 1, 1, 1, 1, 9], 25)
# the belief becomes wrong: 130-tick time-outs with auto-reset inside a run of short calls (a time-out adds a solve tick the plan does not know of)
g, o = run(hk.make_config(E, 4, jitter_seed=5, laps=1, max_episode_steps=130), [20] * 5 + [1] * 40 + [20] * 4 + [3] * 20, 9)
# a partial reset drops the belief; a reset of every env brings it back
g.reset([3, 17, 60], 4); o.reset([3, 17, 60], 4)
for n in (1, 1, 20, 2):
    g.step(n); o.step(n)
cmp(g, o, -1)
g.reset(); o.reset()
for n in (20, 1, 1, 1, 1, 20):
    g.step(n); o.step(n)
cmp(g, o, -2)
# the FOLDED path with a belief that becomes wrong (round 6): past BULK_TICKS a call shorter than the split threshold arms inside its first tick launch, and
# one-lap races end (and auto-reset, out of phase with the belief) inside a long run of short calls that nobody looks at.  An env parked at a solve tick the
# plan did not foresee must keep every call's ticks (hk_env_run.h: the arming is stored for envs that do not enter the loop).
g, o = run(hk.make_config(E, 4, jitter_seed=11, laps=1), [600, 330] + [1] * 150 + [3] * 60 + [2] * 40, 10 ** 6)
assert int(g.env_state()["episodes_done"].sum()) > 0, "no race ended inside the run of short calls: the scenario does not reach the path"
print("optimistic ok")
"""

MODES = {"default": {}, "skew1": {"HK_OPTIMISTIC_SKEW": "1"}, "skew2": {"HK_OPTIMISTIC_SKEW": "2"}, "skew3": {"HK_OPTIMISTIC_SKEW": "3"},
         "off": {"HK_NO_OPTIMISTIC": "1"}, "split": {"HK_SPLIT": "1"}, "split_skew2": {"HK_SPLIT": "1", "HK_OPTIMISTIC_SKEW": "2"},
         # one stream always: every fixed-round call is folded (arms inside its first tick launch, its last tick launch is the guard)
         "fold": {"HK_SPLIT": "0"}, "fold_skew1": {"HK_SPLIT": "0", "HK_OPTIMISTIC_SKEW": "1"}, "fold_skew2": {"HK_SPLIT": "0", "HK_OPTIMISTIC_SKEW": "2"},
         "fold_skew3": {"HK_SPLIT": "0", "HK_OPTIMISTIC_SKEW": "3"}}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_short_calls_match_the_oracle_whatever_the_plan_believes(mode):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HK_") or k in ("HK_LIB_PATH",)}
    env.update(MODES[mode])
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "optimistic ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
