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
    return "tick" if "env_run_kernel" in n else "b1" if "env_b1" in n else "lqn" if "lqn_round" in n else n.split("(")[0].split("::")[-1][:22]
# the timed pass is the FIRST pass: find the first reset, then the pre-roll, then warm-up (5 ticks), then the 20-tick call: print the kernels between the
# second and the third env_arm / env_check after the pre-roll... simply: print every kernel whose start lies within 3 ms after the first 'env_check' that follows > 200 tick launches
cnt = 0; t_mark = None
for k, r in enumerate(rows):
    if nm(r) == "tick": cnt += 1
    if cnt > 250 and nm(r) in ("env_check_kernel",) and t_mark is None: t_mark = int(r["End_Timestamp"]); break
sel = [r for r in rows if t_mark is not None and t_mark <= int(r["Start_Timestamp"]) <= t_mark + 4000000]
t0 = int(sel[0]["Start_Timestamp"]) if sel else 0
for r in sel[:90]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-24s q%-2s start %8.1f us  dur %6.1f us  grid %s" % (nm(r), r.get("Queue_Id"), s / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
