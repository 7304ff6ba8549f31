#!/bin/bash
# timeline of the driver's window: one hk_step(20) from tick 517 (rocprofv3 --kernel-trace of python3 bench.py --steps 20 --warmup 5)
export TMPDIR=/tmp
O=gpurun_out/w20_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv, glob, json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print("value", round(d["value"] / 1e6, 1), "ms", d["ms_per_step"] * 20)
rows = list(csv.DictReader(open(glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = r["Kernel_Name"]
    return "tick" if "env_run_kernel" in n else "b1" if "env_b1" in n else "lqn" if "lqn_" in n else n.split("(")[0].split("::")[-1][:22]
# the timed pass is the FIRST pass: reset, pre-roll (one 512-tick call), warm-up, then the 20-tick call = everything from the end of the kernel before the call's
# first launch (an env_arm_kernel, or — since round 6 folds the arming into the first tick launches — the first tick launch after the warm-up's last) to its last launch
ticks = [k for k, r in enumerate(rows) if nm(r) == "tick"]
# the pre-roll issues >= 250 tick launches; the warm-up (5 ticks) 2 - 4 more; the timed call begins with the next launch after a pause of the stream (host sync): find the first
# gap > 150 us between consecutive kernels after 250 tick launches have gone by, twice (pre-roll -> warm-up is back to back: no gap; warm-up -> timed call: hk_synchronize + barrier)
seen = 0; start = None
for k in range(1, len(rows)):
    if nm(rows[k - 1]) == "tick": seen += 1
    half = str(int(d["config"]["envs_per_gpu"]) * 4 // 2)          # the timed 20-tick call runs as two halves (the 5-tick warm-up as one batch)
    if seen >= 250 and int(rows[k]["Start_Timestamp"]) - int(rows[k - 1]["End_Timestamp"]) > 150000 and (nm(rows[k]) == "env_arm_kernel" or (nm(rows[k]) == "tick" and rows[k].get("Grid_Size_X", rows[k].get("Grid_Size")) == half)):
        start = k; break
if start is None: raise SystemExit("timed call not found")
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 60]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s > 1500000: break
    print("%-24s q%-2s start %8.1f us  end %8.1f  dur %6.1f us  grid %s" % (nm(r), r.get("Queue_Id"), s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
