for v in "" "HK_SPLIT=1" "HK_LAZY_MIN_TICKS=16" "HK_LAZY_MIN_TICKS=8" ; do
  for rep in 1 2; do
  env $v python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', round(d['value']/1e6,1), d['ms_per_step']*20, r.get('launches'), {k:round(v,3) for k,v in r['kernel_total_ms'].items() if v})"
  done
done
