#!/bin/bash
# throughput against the batch size (block-generation quantisation): tools/experiments/ab_size.sh "<env settings or ->" E1 E2 ...
set -o pipefail
e=""; [ "$1" != "-" ] && e="$1"; shift
for n in "$@"; do
  a=$(env $e python bench.py --envs-per-gpu $n --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,1), r['launches'], {k: round(x,1) for k,x in r['kernel_total_ms'].items() if x})") || exit 1
  echo "[$e] E=$n protocol $a"
done
