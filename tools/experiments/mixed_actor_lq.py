"""A 2v2 Oval race with one team on the RL actor (device policy, DecisionPeriod 2) and the other on the LQ game: the handle type that steps in
decision chunks AND solves LQ games.  HK_FISSION=0: the fused kernel for it (HK_NO_FISSION_CHUNKS until round 6).  usage (GPU box): python tools/experiments/mixed_actor_lq.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.policy import Policy
E = 32768
env = hk.RacingEnv(hk.make_config(E, 4, low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_RL, _lib.HK_LOW_LQR, _lib.HK_LOW_LQR], jitter_seed=0x5EED0000))
env.attach_policy(Policy.random(env.obs_dim * 4, 256, 3, seed=101), [0, 1], 2)
env.reset(); env.step(256); env.synchronize()
t0 = time.perf_counter(); env.step(800); env.synchronize(); dt = time.perf_counter() - t0
print("mixed actor + LQ, %d envs: %.1f M env-steps/s" % (E, E * 800 / dt / 1e6))
