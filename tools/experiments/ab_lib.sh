#!/bin/bash
# same-box A/B of library variants and environment switches: each argument is "<variant or ->[:VAR=value ...]" (variant = build/libhk_<variant>.so, "-" = the product)
#   tools/experiments/ab_lib.sh -  noprio  "-:HK_INWAVE=0"      -> protocol window (value, stage totals) and the driver's window (value, median of the repeats), REPS times alternating
set -o pipefail
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    lib=${v%%:*}; e=""; [[ "$v" == *:* ]] && e="${v#*:}"
    [ "$lib" != "-" ] && e="$e HK_LIB_PATH=$PWD/build/libhk_$lib.so"
    a=$(env $e python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1), {k: round(x,1) for k,x in r['kernel_total_ms'].items() if x})") || exit 1
    b=$(env $e python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['window_repeats']['median']/1e6,1))") || exit 1
    echo "[$v] protocol $a | driver window, median of repeats $b"
  done
done
