#!/usr/bin/env python3
"""Reward-shaped / Training-mode handles on the fission schedule against the fused kernel (HK_FISSION=0; HK_NO_FISSION_SHAPED until round 6): env-steps/s of three set-ups (run on the GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
E = 32768
def run(name, **kw):
    g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000, **kw))
    g.reset(); g.step(256); g.synchronize()
    t0 = time.perf_counter(); g.step(1024); g.synchronize(); dt = time.perf_counter() - t0
    print("%-44s %7.1f M env-steps/s" % (name, E * 1024 / dt / 1e6)); g.close()
run("LQNG + rewards", rewards=1)
run("Training mode, LQNG, rewards", env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], rewards=1)
run("planner + LQNG + rewards (16 iterations)", rewards=1, high_mode=[_lib.HK_HIGH_MCTS] * 4, tree_search_depth=8, mcts_iterations=16)
