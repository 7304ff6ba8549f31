"""how many solver passes a wave of the B1 kernel runs in the driver's window (hk_prof_games words 0 / 1: passes, waves with games), by schedule history"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
E = 65536
g = hk.RacingEnv(hk.make_config(E, 4, jitter_seed=0x5EED0000))
g.reset(); g.step(512); g.synchronize(); g.step(5); g.synchronize()
g.prof_enable(True); g.prof_reset()
g.step(20); g.synchronize()
raw = (C.c_int64 * 9)()
g._ck(g.L.hk_prof_games(g.h, raw))
print(os.environ.get("HK_INWAVE", "default"), "passes", raw[0], "waves with games", raw[1], "games", {n: raw[n] for n in (2, 3, 4)}, "passes per busy wave %.2f" % (raw[0] / max(raw[1], 1)),
      g.schedule_info()["multi_player_games"][:40])
