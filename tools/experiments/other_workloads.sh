#!/bin/bash
# the other workloads of bench.py with the arguments of rounds 2 / 3 (profiles/r0N_*_bench_{mcts,mctsrl,a8,rl,a2,second_episodes}.json)
O=gpurun_out/wl; mkdir -p $O
python bench.py --workload mcts --steps 800 --warmup 256 --no-cpu-baseline > $O/bench_mcts.json 2> $O/err.log || exit 1
python bench.py --workload mcts --steps 2000 --warmup 256 --no-cpu-baseline > $O/bench_mcts_2000.json 2>> $O/err.log || exit 1
python bench.py --workload mctsrl --steps 800 --warmup 256 --no-cpu-baseline > $O/bench_mctsrl.json 2>> $O/err.log || exit 1
python bench.py --workload a8 --steps 400 --warmup 200 --no-cpu-baseline > $O/bench_a8.json 2>> $O/err.log || exit 1
python bench.py --workload rl --no-cpu-baseline > $O/bench_rl.json 2>> $O/err.log || exit 1
python bench.py --agents 2 --no-cpu-baseline --no-secondary > $O/bench_a2.json 2>> $O/err.log || exit 1
python bench.py --steps 6000 --warmup 4500 --preroll 0 --no-cpu-baseline --no-secondary > $O/bench_second_episodes.json 2>> $O/err.log || exit 1
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']/1e6,1))
PY
