O=gpurun_out/ab3; mkdir -p $O
for i in 1 2 3; do
  for m in d 1 0; do
    if [ $m = d ]; then python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/$m$i.json 2>>$O/err.log; else HK_INWAVE=$m python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/$m$i.json 2>>$O/err.log; fi
  done
done
python - <<PY
import json
for m in "d10":
    for i in (1,2,3):
        d=json.loads(open("$O/%s%d.json"%(m,i)).read().strip().splitlines()[-1]); r=d["roofline"]
        print(m, i, round(d["value"]/1e6), round(d["window_repeats"]["median"]/1e6), {k:round(v*1e3,1) for k,v in r["kernel_avg_ms"].items() if v}, d["config"]["schedule"]["games_meter"], d["config"]["schedule"]["multi_player_games"][:30])
PY
