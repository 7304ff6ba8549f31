"""Where does a lone, synchronised hk_step(n) lose time against back-to-back calls?  (round 3)
usage: python tools/experiments/short_call_breakdown.py   (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hierarchicalkarting_amd as hk

E = 65536
env = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
for prof in (False, True):
    for n in (1, 2, 4, 8, 20, 40, 100):
        env.prof_enable(False)
        env.reset(); env.step(512); env.synchronize()          # every row starts from the steady state of a fresh race
        env.prof_enable(prof)
        lone = []
        for _ in range(12):
            env.synchronize()
            t0 = time.perf_counter(); env.step(n); t1 = time.perf_counter(); env.synchronize(); t2 = time.perf_counter()
            lone.append((t1 - t0, t2 - t0))
        lone = np.array(lone[2:]) * 1e6
        if n == 20:
            print("   n = 20 samples (us):", " ".join("%.0f" % v for v in lone[:, 1]))
        env.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            env.step(n)
        env.synchronize()
        b2b = (time.perf_counter() - t0) / 10 * 1e6
        print("prof %d  n %3d  lone: issue %7.1f us  total %7.1f us (min %7.1f)   back-to-back %7.1f us/call   ideal %7.1f   lone rate %6.1f M" % (
            prof, n, np.median(lone[:, 0]), np.median(lone[:, 1]), lone[:, 1].min(), b2b, n * 53.0, E * n / np.median(lone[:, 1])), flush=True)
    if prof:
        env.prof_reset()
