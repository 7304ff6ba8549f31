#!/usr/bin/env python3
"""The driver's window (`bench.py --steps 20 --warmup 5`: one hk_step(20) from tick 517) without bench.py around it:
    python tools/short_call.py [--prof] [--ticks 20] [--reps 5]
prints the wall time of each repetition (fresh handle state each: pre-roll to 517 first).  Under
`rocprofv3 --kernel-trace` the last call's kernels show the anatomy of a short call (launch gaps, tail rounds, regroup)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hierarchicalkarting_amd as hk

ap = argparse.ArgumentParser()
ap.add_argument("--prof", action="store_true", help="with hk_prof events, as bench.py runs")
ap.add_argument("--ticks", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--nosync", action="store_true", help="no hk_synchronize between the calls (a host that steps tick by tick without looking): one sync at the end")
ap.add_argument("--sched", type=int, default=-1, help="hipSetDeviceFlags value before the handle exists (1 spin, 2 yield, 4 blocking sync)")
a = ap.parse_args()
if a.sched >= 0:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(a.sched)))
env = hk.RacingEnv(hk.make_config(a.envs, 4, jitter_seed=0x5EED0000))
env.reset()
env.step(517)
env.synchronize()
if a.prof:
    env.prof_enable(True)
out = []
if a.nosync:
    env.synchronize()
    t0 = time.perf_counter()
    for r in range(a.reps):
        env.step(a.ticks)
    env.synchronize()
    dt = time.perf_counter() - t0
    print("ticks %d x %d without a look in between: %.1f us per call (%.0f M env-steps/s)" % (a.ticks, a.reps, dt / a.reps * 1e6, a.envs * a.ticks * a.reps / dt / 1e6))
    sys.exit(0)
for r in range(a.reps):
    env.synchronize()
    t0 = time.perf_counter()
    env.step(a.ticks)
    t1 = time.perf_counter()
    env.synchronize()
    dt = time.perf_counter() - t0
    out.append((dt, t1 - t0))
print("ticks %d prof %d: " % (a.ticks, a.prof) + " ".join("%.3f ms (%.0f M; hk_step returned after %.3f)" % (x * 1e3, a.envs * a.ticks / x / 1e6, i * 1e3) for x, i in out))
