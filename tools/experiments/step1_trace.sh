#!/bin/bash
# kernel timeline of a host that steps tick by tick (hk_step(1) x 40 from tick 517, no look in between): rocprofv3 --kernel-trace of tools/short_call.py
export TMPDIR=/tmp
O=gpurun_out/s1_kt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/short_call.py --ticks 1 --reps 40 --nosync > $O/out.log 2> $O/err.log
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = r["Kernel_Name"]
    return "tick" if "env_run_kernel" in n else "b1" if "env_b1" in n else "lqn" if "lqn_" in n else n.split("(")[0].split("::")[-1][:22]
tail = rows[-70:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-22s q%-2s start %8.1f us  end %8.1f  dur %6.1f us  grid %s" % (nm(r), r.get("Queue_Id"), s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
