#!/usr/bin/env python3
"""Where the tick kernel's and the B1 kernel's vector instructions go: SQ_INSTS_VALU / SQ_THREAD_CYCLES_VALU of 16 steady-state ticks with one region
compiled out at a time (-DHK_DUMMY_NO_*: timing builds of tools/build_variant.py, never the product).  The state is the PRODUCT's after 512
ticks (dumped once, loaded into every variant with hk_set_*_state), so every variant starts from the same race and has 16 ticks to drift.

  python tools/experiments/region_cost.py dump <file.npz>            (default library)
  HK_LIB_PATH=build/libhk_<variant>.so rocprofv3 --pmc ... -- python3 tools/experiments/region_cost.py run <file.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import hierarchicalkarting_amd as hk

cfg = dict(num_envs=65536, num_agents=4, jitter_seed=0x5EED0000)
if sys.argv[1] == "dump":
    env = hk.RacingEnv(hk.make_config(**cfg)); env.reset(); env.step(512); env.synchronize()
    np.savez(sys.argv[2], agents=env.agent_state(), envs=env.env_state())
else:
    d = np.load(sys.argv[2])
    env = hk.RacingEnv(hk.make_config(**cfg)); env.reset()
    env.set_agent_state(d["agents"]); env.set_env_state(d["envs"])
    env.step(16); env.synchronize()
    print("ran 16 ticks")
