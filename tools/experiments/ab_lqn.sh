#!/bin/bash
# same-box A/B of the solver launch of a spread field (round 6): HK_LQN=pair (the pair / matrix-core kernel, the schedule before round 6) against the
# spread solver's launch (hk_lq_spread.h) and the default (in-wave solves in env_b1_kernel), on the driver's window and the protocol window, alternating, three times each
O=gpurun_out/ab_lqn; mkdir -p $O
for rep in 1 2 3; do
  for mode in ${MODES:-pair spread inwave}; do
    unset HK_LQN HK_INWAVE
    if [ $mode = pair ]; then export HK_LQN=pair HK_INWAVE=0; fi      # the schedule before round 6
    if [ $mode = spread ]; then export HK_INWAVE=0; fi                 # queues + the spread solver's launch
    # inwave: the default (env_b1_kernel solves its own games while the meter says the field has spread)
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/drv_${mode}_$rep.json 2>> $O/err.log || exit 1
    python bench.py --no-cpu-baseline --no-secondary > $O/proto_${mode}_$rep.json 2>> $O/err.log || exit 1
  done
done
unset HK_LQN HK_INWAVE
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(f.split('/')[-1], round(d['value']/1e6,1), {k: round(v,2) for k,v in (r.get("kernel_total_ms") or {}).items() if v}, d.get("window_repeats",{}).get("median"))
PY
