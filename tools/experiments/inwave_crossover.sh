#!/bin/bash
# where the in-wave solves stop paying: the driver's kind of window (20 ticks after 5 warm-up ticks) at several points of the race, i.e. at several
# numbers of multi-player games per B1 launch, with the games solved in-wave (HK_INWAVE=1) and by a solver launch (HK_INWAVE=0)
O=gpurun_out/cross; mkdir -p $O
for pre in ${PREROLLS:-512 576 640 768 1024 1536 2560}; do
  for m in 1 0; do
    for i in 1 2 3; do
      HK_INWAVE=$m python bench.py --gpus 1 --steps 20 --warmup 5 --preroll $pre --no-secondary --no-cpu-baseline > $O/p${pre}_m${m}_$i.json 2>> $O/err.log || exit 1
    done
  done
done
python - <<PY
import json
for pre in [int(x) for x in "${PREROLLS:-512 576 640 768 1024 1536 2560}".split()]:
    row = []
    for m in (1, 0):
        v = []; g = 0
        for i in (1, 2, 3):
            d = json.loads(open("$O/p%d_m%d_%d.json" % (pre, m, i)).read().strip().splitlines()[-1])
            v.append(d["value"] / 1e6); r = d["roofline"]
            b1 = r["kernel_avg_ms"].get("env_b1_kernel", 0) * 1e3
            nb1 = r["kernel_total_ms"].get("env_b1_kernel", 0) * 1e3 / max(b1, 1e-9)
            g = sum(r.get("multi_player_games_solved", {}).values()) / max(nb1, 1)
        row.append("%s  b1 %.0f us" % (" ".join("%6.0f" % x for x in sorted(v)), b1))
    print("tick %4d  games/half-launch ~%4.0f   in-wave: %s   |  launch: %s" % (pre + 5, g, row[0], row[1]))
PY
