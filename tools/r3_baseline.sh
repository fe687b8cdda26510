#!/bin/bash
# Baseline figures at the start of a round: graph probe, the data-parallel rehearsal on one GPU, the default bench line.
set -e
out=gpurun_out/r3_baseline
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_probe tools/graph_probe.hip
timeout -k 10 120 /tmp/graph_probe > $out/graph_probe.txt 2>&1
for ex in sum shard gather; do
  for n in 2 8; do
    KALDI_AMD_DP_EXCHANGE=$ex KALDI_BENCH_LOOPBACK_DP=$n timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/dp_${ex}_$n.json 2> $out/dp_${ex}_$n.err
    echo "$ex $n done"
  done
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
echo ok
