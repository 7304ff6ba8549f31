"""The tick kernel must not depend on how its loop is written.

Round 2 recorded two cases in which a logically identical rewrite of env_run_kernel's loop changed results: the loop end as an
if / else-if / else chain made the Training-mode instantiation fail its parity test, and the 8-lane build "decoded a wrong
final_steer" on the wave-uniform loop and was pinned to an older loop.  Both forms are built here — the default library and a variant
with -DHK_LOOP_IFELSE (the chain form, hk_env_run.h) for both lane-group widths — and the Training-mode, 8-agent and headline parity
runs must pass on each, bit for bit against the oracle.  The variant is compiled on this box when build/libhk_ifelse.so is missing."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "libhk_ifelse.so")

CHILD = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
assert os.path.samefile(_lib.LIB_PATH, %r)

def cmp(g, o, t):
    gs, os_ = g.agent_state(), o.agent_state()
    for name in gs.dtype.names:
        x, y = gs[name], os_[name]
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        assert np.array_equal(x, y), (t, name, np.argwhere(x != y)[:3].tolist())

def run(b, calls):
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    t = 0
    for n in calls:
        g.step(n); o.step(n); t += n
        cmp(g, o, t)

# the Training-mode instantiation <true, true, true> (round 2: failed with the chain form), rewards on, Complex track, time-outs
run(hk.make_config(24, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=1, jitter_seed=0, track="complex"),
    (100, 1, 199, 57, 243, 300))
# the headline instantiation: long (eager, 12 ticks per launch) and short calls, auto-reset
run(hk.make_config(512, 4, jitter_seed=0x5EED0000, laps=1), (130, 70, 20, 7, 1, 1, 2, 300, 900))
# 8 lanes per env (round 2: pinned to the older loop)
run(hk.make_config(96, 8, jitter_seed=0x5EED0000, laps=1), (130, 70, 20, 7, 1, 300))
run(hk.make_config(16, 8, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1] * 4 + [0] * 4, laps=1, max_episode_steps=250, rewards=1, jitter_seed=0), (100, 151, 120))
print("loop form ok")
"""


def _build_variant():
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(os.path.join(ROOT, "hierarchicalkarting_amd", "csrc", f))
                                                              for f in os.listdir(os.path.join(ROOT, "hierarchicalkarting_amd", "csrc"))):
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "ifelse", "-DHK_LOOP_IFELSE=1", "--all"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("variant", ["default", "ifelse"])
def test_parity_does_not_depend_on_the_loop_form(variant):
    """Round 3 found what made round 2's builds differ: not the loop, but where the register allocator of this toolchain put a VGPR spill
    store — in the join block of a divergent branch, ahead of the instruction that restores EXEC, so that the lanes of the other side
    reload a stale value (in the chain form of the current sources: m_FinalStats.Steer of the Training-mode
    instantiation, four ticks old, for the karts faster than 5 m/s).  tools/check_spill_exec.py finds such stores in the listing; the
    product build recompiles a flagged unit with the next of its result-neutral variants and refuses to ship one no variant cleans (__graft_entry__.build).  Here: a build the guard calls clean must be bit-identical to the
    oracle; a variant it flags may fail — and a variant that fails must have been flagged (no unexplained difference)."""
    import json
    env = dict(os.environ)
    lib = os.path.join(ROOT, "hierarchicalkarting_amd", "libhk.so")
    flagged = []
    if variant == "ifelse":
        _build_variant()
        lib = LIB
        env["HK_LIB_PATH"] = LIB
        flagged = json.load(open(LIB[:-3] + ".guard.json"))["spill_stores_ahead_of_exec_restore"]
    else:
        env.pop("HK_LIB_PATH", None)
        rec = json.load(open(os.path.join(ROOT, "build", "obj", "codegen_guard.json")))       # written by __graft_entry__.build()
        assert rec and all(not v["findings"] for v in rec.values()), "the product library was built without a clean code-generation guard: %s" % rec
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, lib)], env=env, capture_output=True, text=True, timeout=900)
    ok = r.returncode == 0 and "loop form ok" in r.stdout
    if flagged:
        print("variant flagged by the guard (%d store(s)); parity %s" % (len(flagged), "held" if ok else "failed, as it may"))
        return
    assert ok, r.stdout[-2000:] + r.stderr[-4000:]
