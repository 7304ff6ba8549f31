#!/usr/bin/env python3
"""How persistent are multi-player games per env?  (run on the GPU box)  Steps the headline workload one solve cadence at a time from tick 512 and
reads the scheduling hint of every env (reserved[1] & 16: the env's last solve tick queued a multi-player game)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hierarchicalkarting_amd as hk
E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 768
b = hk.make_config(E, 4, jitter_seed=0x5EED0000, auto_reset=1)
g = hk.RacingEnv(b)
g.reset(); g.step(512)
cnt = np.zeros(E, np.int32); runs = []; cur = np.zeros(E, np.int32); per = []; new = []
prev = np.zeros(E, bool)
for k in range(N):
    g.step(4)
    hint = (g.env_state()["reserved"][:, 1] & 16) != 0
    cnt += hint
    per.append(int(hint.sum())); new.append(int((hint & ~prev).sum()))
    ended = ~hint & (cur > 0)
    runs += list(cur[ended]); cur[ended] = 0; cur[hint] += 1
    prev = hint
runs += list(cur[cur > 0])
runs = np.array(runs)
print("envs", E, "cadences", N)
print("envs with a queued game per cadence: mean %.1f  min %d  max %d" % (np.mean(per), min(per), max(per)))
print("envs that START a pack per cadence (hint 0 -> 1): mean %.2f  total %d" % (np.mean(new), sum(new)))
print("envs that ever queued: %d (%.2f %%); their cadences with a game: mean %.1f, median %d, p90 %d, max %d" % (
    (cnt > 0).sum(), 100.0 * (cnt > 0).mean(), cnt[cnt > 0].mean(), np.median(cnt[cnt > 0]), np.percentile(cnt[cnt > 0], 90), cnt.max()))
print("runs of consecutive cadences with a game: %d runs, mean length %.1f, median %d, p90 %d, max %d; share of game-cadences in runs >= 8: %.2f" % (
    len(runs), runs.mean(), np.median(runs), np.percentile(runs, 90), runs.max(), runs[runs >= 8].sum() / max(runs.sum(), 1)))
