"""the race start call by call (hk_schedule_info + hk_prof stage totals per 32-tick call from a reset of every env): where the first 512 ticks go"""
import sys, os
sys.path.insert(0, os.getcwd())
import hierarchicalkarting_amd as hk
E = 65536
g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
g.reset(); g.synchronize()
g.prof_enable(True)
import time
t = 0
for k in range(18):
    g.prof_reset()
    t0 = time.perf_counter()
    g.step(32); g.synchronize(); t += 32
    dt = time.perf_counter() - t0
    s = g.schedule_info(); games = g.prof_games(); ms = g.prof_read()
    print("%4d  %6.2f ms  %-6s %-30s" % (t, dt * 1e3, s["games_meter"], s["multi_player_games"][:30]), {n: v for n, v in games.items() if v}, {k: (round(v[0], 2), v[1]) for k, v in ms.items() if v[1]})
