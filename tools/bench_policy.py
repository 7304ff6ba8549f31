#!/usr/bin/env python3
"""Time policy_mlp_kernel alone (HIP events around the launch, via hk_prof) inside the decision loop of a 4-agent env:
rows per launch = E * 4 (one actor drives every agent)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.policy import Policy

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
out = {}
for (A, stack, hidden, layers) in ((4, 4, 256, 3), (4, 4, 128, 3), (4, 8, 256, 3)):
    env = hk.RacingEnv(hk.make_config(E, A, low_mode=[_lib.HK_LOW_RL] * A, jitter_seed=1))
    in_dim = env.obs_dim * stack
    pol = Policy.random(in_dim, hidden, layers, stack=stack, seed=1)
    env.attach_policy(pol, list(range(A)), 2)
    env.reset()
    env.step(100)
    env.prof_enable(True); env.prof_reset()
    env.step(200)
    pr = env.prof_read()
    ms, n = pr["policy_mlp_kernel"]
    rows = E * A
    flop = rows * 2.0 * (in_dim * hidden + (layers - 1) * hidden * hidden + 4 * hidden)
    out["%d->%dx%d" % (in_dim, hidden, layers)] = {"ms": round(ms / n, 4), "tflops": round(flop / (ms / n * 1e-3) / 1e12, 1),
                                                   "obs_ms": round(pr["observe+stack"][0] / max(pr["observe+stack"][1], 1), 4)}
    env.close()
print(os.path.basename(os.environ.get("HK_LIB_PATH", "default")), json.dumps(out))
