set -e
export TMPDIR=/tmp
for v in fis nofis; do
  if [ $v = nofis ]; then export HK_FISSION=0; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/step1_$v -o t --output-format csv -- python3 tools/experiments/step1_trace.py 1 > gpurun_out/step1_$v.log 2>&1
  tail -1 gpurun_out/step1_$v.log
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/step1_$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("   %-60s calls %6s avg %8.1f us total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
