#!/bin/bash
# the split batch as 2 / 3 / 4 parts on as many streams: headline window and the 20-tick call
for w in 2 3 4 2 3 4; do
  HK_SPLIT_WAYS=$w python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ways $w', round(d['value']/1e6,1))" || exit 1
done
for w in 2 3 4; do
  HK_SPLIT_WAYS=$w python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ways $w, 20 ticks', round(d['value']/1e6,1))" || exit 1
done
