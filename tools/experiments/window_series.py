#!/usr/bin/env python3
"""the driver's window as a series: value of every 20-tick window from tick 517 on, in order (is the first one special?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
E = 65536
g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
for trial in range(3):
    g.reset(); g.step(512); g.step(5); g.synchronize()
    out = []
    for k in range(12):
        t0 = time.perf_counter(); g.step(20); g.synchronize(); dt = time.perf_counter() - t0
        out.append(round(E * 20 / dt / 1e6))
    print("trial", trial, out)
