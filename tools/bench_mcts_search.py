#!/usr/bin/env python3
"""Latency of one planner search launch (hk_reset = env_reset_kernel + mcts_search_kernel + sync) vs iterations / batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
MC = _lib.HK_HIGH_MCTS
CASES = ((16, 4, 16), (16, 4, 64), (16, 4, 256), (4096, 4, 64), (16, 2, 64), (16, 1, 64), (16384, 4, 64), (32768, 4, 64), (65536, 4, 64))
for E, A, it in CASES:
    env = hk.RacingEnv(hk.make_config(E, A, track="complex", high_mode=[MC] * A, tree_search_depth=8, mcts_iterations=it,
                                      mcts_initial_iterations=it, jitter_seed=3))
    env.reset(); env.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        env.reset(); env.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("E %5d A %d iterations %4d: %8.2f ms per launch, %7.3f ms per iteration" % (E, A, it, dt * 1e3, dt * 1e3 / it), flush=True)
    env.close()
