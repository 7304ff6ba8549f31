"""First tick / field / agent at which a library differs from the oracle on one long call (scheduling kept: a fresh handle per probe, one hk_step(k)).
usage: HK_LIB_PATH=build/libhk_<variant>.so python tools/experiments/first_divergence.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib

def cfg():
    return hk.make_config(24, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=1, jitter_seed=0, track="complex")

def diff(k):
    b = cfg()
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    g.step(k); o.step(k)
    gs, os_ = g.agent_state(), o.agent_state()
    bad = []
    for name in gs.dtype.names:
        x, y = np.ascontiguousarray(gs[name]), np.ascontiguousarray(os_[name])
        if x.dtype.kind == "f":
            x = x.view(np.uint32); y = y.view(np.uint32)
        if not np.array_equal(x, y):
            idx = np.argwhere((x != y) if x.ndim == 2 else (x != y).any(axis=tuple(range(2, x.ndim))))
            bad.append((name, idx[:4].tolist(), len(idx)))
    return bad, gs, os_

lo, hi = 0, 100
assert diff(hi)[0], "no difference at 100 ticks"
while hi - lo > 1:
    mid = (lo + hi) // 2
    if diff(mid)[0]: hi = mid
    else: lo = mid
bad, gs, os_ = diff(hi)
print("first differing call length:", hi)
for name, idx, n in bad:
    e, a = idx[0]
    print("  %-24s %3d mismatches, first env %d agent %d: lib %r oracle %r" % (name, n, e, a, gs[name][e, a], os_[name][e, a]))

# the LQ taps of the differing egos at that tick
b = cfg(); b.cfg.debug_taps = 1
g = hk.RacingEnv(b); o = O.OracleEnv(b)
g.reset(); o.reset(); g.step(hi); o.step(hi)
gs, os_ = g.agent_state(), o.agent_state()
d = np.argwhere(gs["steering"].view(np.uint32) != os_["steering"].view(np.uint32))
print("with taps on: %d egos differ in steering" % len(d))
for e, a in d[:6]:
    gd, od = g.lq_debug(int(e), int(a)), o.lq_debug(int(e), int(a))
    print(" env %d ego %d: n_players lib %d oracle %d  players %s / %s  branch %s / %s" % (e, a, gd.n_players, od.n_players, list(gd.player_agent)[:4], list(od.player_agent)[:4], list(gd.branch)[:4], list(od.branch)[:4]))
    for i in range(max(gd.n_players, 1)):
        for nm in ("initial", "target", "target_w"):
            x, y = list(getattr(gd, nm)[i]), list(getattr(od, nm)[i])
            if x != y: print("    player %d %-8s lib %r\n                      ora %r" % (i, nm, x, y))
    print("    u0 lib %r oracle %r" % (list(gd.u0), list(od.u0)))
