# A/B of kernel variants on the workloads that run the FUSED tick kernel: bash tools/experiments/ab_wl.sh base <variant> ...
for v in "$@"; do
  export HK_LIB_PATH=$PWD/build/libhk_$v.so
  a=$(python bench.py --agents 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['value']/1e6,1))")
  b=$(python bench.py --workload rl --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_total_ms']['env_run_kernel'],1))")
  c=$(python bench.py --workload mcts --steps 800 --warmup 256 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['roofline']['kernel_total_ms']['env_run_kernel'],1))")
  echo "$v: a2 $a | rl $b | mcts $c"
done
