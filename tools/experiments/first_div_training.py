import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
import hierarchicalkarting_amd as hk
from hierarchicalkarting_amd import _lib
from hierarchicalkarting_amd.policy import Policy
b = hk.make_config(12, 2, env_mode=_lib.HK_MODE_TRAINING, training_agents=[1, 0], rewards=1, high_mode=[_lib.HK_HIGH_MCTS, _lib.HK_HIGH_MCTS],
                   low_mode=[_lib.HK_LOW_RL, _lib.HK_LOW_LQR], tree_search_depth=8, mcts_iterations=10, laps=1, max_episode_steps=350, jitter_seed=0)
g = hk.RacingEnv(b); o = O.OracleEnv(b)
pol = Policy.random(g.obs_dim * 4, 64, 2, seed=4)
g.attach_policy(pol, [0], 2); o.attach_policy(pol, [0], 2)
g.reset(); o.reset()
for t in range(1, 121):
    g.step(1); o.step(1)
    gs, os_ = g.agent_state(), o.agent_state()
    bad = [nm for nm in gs.dtype.names if not np.array_equal(np.ascontiguousarray(gs[nm]).view(np.uint8), np.ascontiguousarray(os_[nm]).view(np.uint8))]
    if bad:
        for nm in bad[:6]:
            idx = np.argwhere(gs[nm] != os_[nm])[:3]
            print("tick", t, nm, idx.tolist(), [(gs[nm][tuple(i)], os_[nm][tuple(i)]) for i in idx[:2]])
        e = int(np.argwhere(gs[bad[0]] != os_[bad[0]])[0][0])
        for nm in ("px", "pz", "section_index", "lane", "flags", "trig_lo", "trig_hi", "tele_completed_laps", "tele_total_time", "time_steps"):
            print("   env", e, nm, gs[nm][e].tolist(), os_[nm][e].tolist())
        print("   env words", g.env_state()[e], o.env_state()[e])
        break
else:
    print("no divergence stepping one tick at a time")
