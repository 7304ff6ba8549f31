#!/bin/bash
# same-box A/B of environment switches (run on the GPU box from the repo root): each argument is a string of VAR=value pairs ("-" = defaults)
#   tools/experiments/ab_env.sh - "HK_INWAVE=0" "HK_FISSION=0"      -> protocol window and the driver's 20-tick window per setting
set -o pipefail
for v in "$@"; do
  e=""; [ "$v" != "-" ] && e="$v"
  a=$(env $e python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1), r['launches'], round(r['avg_launch_ms'],4), {k: round(x,1) for k,x in r['kernel_total_ms'].items() if x})") || exit 1
  b=$(env $e python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1))") || exit 1
  echo "[$v] protocol $a | driver-args $b"
done
