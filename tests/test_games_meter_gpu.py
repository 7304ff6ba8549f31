"""The games meter and the lazily joined parts of a split batch across a race start stepped tick by tick (round 6).

Default: every call of a plain handle of >= 8 192 envs runs as two halves on two streams whose parts stay open from call to call (hk_api.hip split_join) —
a host that steps tick by tick keeps both halves' meter words current, reaches the sparse (in-wave) schedule once the field has spread, and sees the same
state, bit for bit, as a host that steps in long calls, whatever schedule each was given; a getter in between joins the parts.

HK_LAZY_JOIN=0 (the schedule before): one-tick calls of a spread field run as ONE batch, so the second half's meter word stops being written after the
race start.  Read for ever with its last value — the start's counts — it would keep such a host on the dense schedule for the rest of the race: the host
only reads the parts the call before ran as, and a part that launches again after a change of shape starts its words over."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os
sys.path.insert(0, %(root)r)
import numpy as np
import hierarchicalkarting_amd as hk

def same(a, r):
    for name in a.dtype.names:
        x, y = a[name], r[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), name

lazy_join = os.environ.get("HK_LAZY_JOIN") != "0"
b = hk.make_config(8192, 4, jitter_seed=5, laps=3, max_episode_steps=4000)
g = hk.RacingEnv(b); ref = hk.RacingEnv(b)
g.reset(); ref.reset()
g.step(1)
first = g.schedule_info()
assert first["streams"] == 2, first                 # the close field of a race start: two halves
for k in range(899):
    g.step(1)
    if k == 300:                                    # a look in the middle of the run joins the open parts and must show the long-call host's state
        ref.step(302)
        same(g.agent_state(), ref.agent_state())
last = g.schedule_info()
assert last["call_ticks"] == 1 and last["streams"] == (2 if lazy_join else 1), last
assert last["games_meter"] == "sparse", last
assert "in-wave" in last["multi_player_games"], last
ref.step(598)
same(g.agent_state(), ref.agent_state())
# and on: a long call after the tick-by-tick stretch (with HK_LAZY_JOIN=0: two halves again, from words started over, not from the race start's)
g.step(64); ref.step(64)
again = g.schedule_info()
assert again["streams"] == 2 and again["games_meter"] in ("sparse", "medium"), again
same(g.agent_state(), ref.agent_state())
# short calls of mixed sizes with results read through the device-pointer path in between (settle_for_pointer joins too)
for n in (1, 3, 1, 7, 2, 1, 1, 20, 1):
    g.step(n); ref.step(n)
assert g.device_results_ptr() != 0
assert np.array_equal(g.episode_results().view(np.uint8), ref.episode_results().view(np.uint8))
same(g.agent_state(), ref.agent_state())
g.close(); ref.close()
print("meter ok")
"""


@pytest.mark.parametrize("mode", ["default", "joined_every_call"])
def test_tick_by_tick_host_after_a_split_start(mode):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HK_") or k in ("HK_LIB_PATH",)}
    if mode == "joined_every_call":
        env["HK_LAZY_JOIN"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "meter ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
