"""what the games meter tells the host, call by call (hk_schedule_info), through a burst of restarts: 65 536 envs, 200-tick calls from tick 3 600"""
import sys, os
sys.path.insert(0, os.getcwd())
import hierarchicalkarting_amd as hk
E = 65536
g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
g.reset(); g.step(3600); g.synchronize()
g.prof_enable(True)
t = 3600
for k in range(24):
    g.prof_reset()
    g.step(200); g.synchronize(); t += 200
    s = g.schedule_info(); games = g.prof_games(); ms = g.prof_read()
    print(t, s["games_meter"], "|", s["multi_player_games"][:28], "|", {n: v for n, v in games.items() if v}, {k: round(v[0], 1) for k, v in ms.items() if v[1]})
