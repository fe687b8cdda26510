#!/bin/bash
# kernel trace + per-kernel stats of one bench workload.  usage: tools/trace_workload.sh <workload>
set -e
ROOT=$(pwd); wl=$1
out=$ROOT/gpurun_out/trace_$wl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $ROOT/bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $out/kt.log 2>&1
cd $ROOT
python3 - $out/kt/kt_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    print("%-100s n=%5s avg=%8.2f us tot=%8.3f ms %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
python3 tools/timeline.py $out/kt/kt_kernel_trace.csv | head -40
