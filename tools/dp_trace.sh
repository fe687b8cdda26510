#!/bin/bash
# Kernel trace of the data-parallel rehearsal (one-rank stand-in) and the timeline of one step.
set -e
ROOT=$(pwd)
out=$ROOT/gpurun_out/r3_dp_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ch in ${CHUNKS:-3 1}; do
  KALDI_AMD_DP_CHUNKS=$ch KALDI_AMD_DP_EXCHANGE=shard KALDI_BENCH_LOOPBACK_DP=8 KALDI_BENCH_LOOPBACK_ONE_RANK=${ONE_RANK-1} timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$ch -o kt -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/kt$ch.log 2>&1
  f=$out/kt$ch/kt_kernel_trace.csv
  python3 $ROOT/tools/timeline.py $f > $out/timeline_$ch.txt
  echo "chunks $ch: $(grep -o '"ms_per_step": [0-9.]*' $out/kt$ch.log | head -1)"
done
