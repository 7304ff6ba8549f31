import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import hierarchicalkarting_amd as hk
E = 8192 + 96
g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=1003, laps=1, max_episode_steps=700))
g.reset()
t = 0
for n in [1, 1, 2, 5, 20, 20, 64, 100, 1, 1, 20, 130, 1, 20, 260, 1, 1, 1, 20, 520, 20, 1]:
    g.step(n); t += n
    s = g.schedule_info()
    print(t, n, s["rounds"], s["streams"], s["games_meter"], s["multi_player_games"][:22], s["optimistic_plan"])
