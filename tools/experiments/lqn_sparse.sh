for b in 1024 256 128 64 32 1024 128; do
  HK_LQN_SPARSE_BLOCKS=$b python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('sparse blocks $b', round(d['value']/1e6,1), {k:round(v,1) for k,v in (r.get('kernel_total_ms') or {}).items() if v})" || exit 1
done
HK_LQN_SPARSE_BLOCKS=128 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 ticks, 128', round(d['value']/1e6,1))"
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20 ticks, 1024', round(d['value']/1e6,1))"
