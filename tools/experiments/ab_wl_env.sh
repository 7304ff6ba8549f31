#!/bin/bash
# same-box A/B of environment switches on one of bench.py's other workloads: tools/experiments/ab_wl_env.sh "<bench args>" - "HK_INWAVE=0" "HK_INWAVE=0 HK_LQN=pair" ...
set -o pipefail
args="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    e=""; [ "$v" != "-" ] && e="$v"
    a=$(env $e python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), {k: round(x,1) for k,x in (d.get('kernel_total_ms') or d.get('roofline',{}).get('kernel_total_ms') or {}).items() if x})") || exit 1
    echo "[$v] $a"
  done
done
