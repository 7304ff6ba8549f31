import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
print("lib", _lib.LIB_PATH)
def first_div(b, n, step=1):
    g = hk.RacingEnv(b); o = O.OracleEnv(b)
    g.reset(); o.reset()
    gs, os_ = g.agent_state(), o.agent_state()
    bad = [nm for nm in gs.dtype.names if not np.array_equal(np.ascontiguousarray(gs[nm]).view(np.uint8), np.ascontiguousarray(os_[nm]).view(np.uint8))]
    print("after reset:", bad)
    for t in range(1, n + 1, step):
        g.step(step); o.step(step)
        gs, os_ = g.agent_state(), o.agent_state()
        bad = [nm for nm in gs.dtype.names if not np.array_equal(np.ascontiguousarray(gs[nm]).view(np.uint8), np.ascontiguousarray(os_[nm]).view(np.uint8))]
        if bad:
            nm = bad[0]
            idx = np.argwhere(gs[nm] != os_[nm])[:4]
            print("tick", t, "fields", bad[:12], "first", nm, idx.tolist(), [ (gs[nm][tuple(i)], os_[nm][tuple(i)]) for i in idx[:2]])
            ge, oe = g.env_state(), o.env_state()
            print("env words equal:", {k: bool(np.array_equal(ge[k], oe[k])) for k in ge.dtype.names})
            return
    print("no divergence in", n)
first_div(hk.make_config(24, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=1, jitter_seed=0, track="complex"), 100)
first_div(hk.make_config(24, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=0, jitter_seed=0, track="complex"), 100)
first_div(hk.make_config(24, 4, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 1, 0, 0], laps=1, max_episode_steps=300, rewards=1, jitter_seed=0), 100)
