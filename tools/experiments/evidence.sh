#!/bin/bash
# the evidence set of a commit (run on the GPU box from the repo root): bench lines, rocprofv3 kernel trace, PMC passes
set -o pipefail
R=$PWD; O=$R/gpurun_out/ev; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_driver_args.json 2>> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprofv3.json 2> $O/kt.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1024 --warmup 0 --no-cpu-baseline --no-secondary > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 1024 --warmup 0 --no-cpu-baseline --no-secondary > $O/pmc_sq1.json 2> $O/pmc_sq1.err || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 1024 --warmup 0 --no-cpu-baseline --no-secondary > $O/pmc_sq2.json 2> $O/pmc_sq2.err || exit 1
cd $R
# fold the PMC passes (HK_COMMIT: the commit the tree was pushed from; there is no .git on the GPU box)
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_summary.json \
  --sq $(find $O/pmc_sq1 -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq2 -name "*counter_collection.csv" | head -1) --last 300 --env-steps-per-launch 131072 --waves-per-simd 3 \
  --commit "${HK_COMMIT:-unknown}" --command "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 1024 --warmup 0 --no-cpu-baseline --no-secondary (one pass each: FETCH_SIZE; WRITE_SIZE; 8 SQ counters; 8 more); default schedule (two halves on two streams): a launch advances 32 768 envs by 4 ticks" > $O/pmc_fold.log 2>&1 || exit 1
python bench.py --workload lqbatch --steps 20 --warmup 3 > $O/bench_lqbatch.json 2>> $O/bench.err
find $O -name "*.csv" | head -20
