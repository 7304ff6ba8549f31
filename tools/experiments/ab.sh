for v in "$@"; do
  HK_LIB_PATH=$PWD/build/libhk_$v.so python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), d['roofline']['launches'], round(d['roofline']['avg_launch_ms'],4))" || exit 1
done
