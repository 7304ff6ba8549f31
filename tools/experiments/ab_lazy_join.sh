#!/bin/bash
# short calls with the parts of a split call joined lazily (HK_LAZY_JOIN=0: joined at the end of every call), one stream (default below 8 ticks) and two (HK_SPLIT=1)
for rep in 1 2; do
for e in "" "HK_LAZY_JOIN=0" "HK_SPLIT=1" "HK_SPLIT=1 HK_LAZY_JOIN=0"; do
  env $e python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', 'protocol', round(d['value'] / 1e6), {k.split('_')[1]: round(v['us_per_call'], 1) for k, v in d['host_driven'].items()})"
  env $e python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('    driver window', round(d['value'] / 1e6), 'median of repeats', round(d['window_repeats']['median'] / 1e6))"
done
done
