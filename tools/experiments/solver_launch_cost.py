"""What does a steady-state solver launch cost, and what does it depend on?  (round 3)
hk_step(64) calls from tick 512: per call the solver stage's time and launches (hk_prof_read) against the games it solved (hk_prof_games)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hierarchicalkarting_amd as hk

env = hk.RacingEnv(hk.make_config(65536, 4, jitter_seed=0x5EED0000))
env.reset(); env.step(512); env.synchronize()
env.prof_enable(True)
rows = []
for k in range(40):
    env.prof_reset()
    env.step(64); env.synchronize()
    p = env.prof_read(); g = env.prof_games()
    lq = [v for kname, v in p.items() if "lqn" in kname][0]
    rows.append((g.get(2, 0), g.get(3, 0), g.get(4, 0), lq[1], lq[0] * 1e3))
    print("call %2d: 2-player %6d  3-player %4d  4-player %3d   solver launches %2d  total %7.1f us  (%.1f us per launch)" % (k, *rows[-1], rows[-1][4] / max(rows[-1][3], 1)), flush=True)
