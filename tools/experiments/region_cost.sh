#!/bin/bash
# Region-cost experiment (see region_cost.py): on the GPU box, from the repo root:   bash tools/experiments/region_cost.sh base notrig noenter ...
# Every name is a build/libhk_<name>.so of tools/build_variant.py ("base" = a variant built with no -D).  One stream (HK_SPLIT=0) so a kernel's
# counters are its own.  Output: gpurun_out/region_cost/<name>/ (rocprofv3 counter csv) and gpurun_out/region_cost/summary.json.
set -e
export HK_SPLIT=0 TMPDIR=/tmp
out=gpurun_out/region_cost; mkdir -p $out
python3 tools/experiments/region_cost.py dump /tmp/rc_state.npz
for v in "$@"; do
  export HK_LIB_PATH=build/libhk_$v.so
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $out/$v -o pmc --output-format csv -- python3 tools/experiments/region_cost.py run /tmp/rc_state.npz > $out/$v.log 2>&1
  echo "done $v"
done
python3 tools/experiments/region_cost_summary.py $out "$@" | tee $out/summary.txt
