#!/bin/bash
# The data-parallel step rehearsed on one GPU (loopback communicator): dist tests, then bench lines per exchange.
set -e
out=gpurun_out/r3_dp
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_dist.py -x -q --timeout 120 --timeout-method thread > $out/pytest_dist.txt 2>&1 || { tail -40 $out/pytest_dist.txt; exit 1; }
tail -3 $out/pytest_dist.txt
for cfg in "shard 8 0" "shard 8 1" "shard 2 0" "sum 8 0"; do
  set -- $cfg
  env_one=""; [ "$3" = "1" ] && env_one="KALDI_BENCH_LOOPBACK_ONE_RANK=1"
  env KALDI_AMD_DP_EXCHANGE=$1 KALDI_BENCH_LOOPBACK_DP=$2 $env_one timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/dp_$1_$2_$3.json 2> $out/dp_$1_$2_$3.err || { tail -20 $out/dp_$1_$2_$3.err; exit 1; }
  python -c "
import json
d=json.load(open('$out/dp_$1_$2_$3.json'))
print('$cfg', d['ms_per_step'], d['value'], d['check'])"
done
for ch in 1 2 8; do
  KALDI_AMD_DP_CHUNKS=$ch KALDI_AMD_DP_EXCHANGE=shard KALDI_BENCH_LOOPBACK_DP=8 KALDI_BENCH_LOOPBACK_ONE_RANK=1 timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/dp_chunks_$ch.json 2> $out/dp_chunks_$ch.err
  python -c "
import json
d=json.load(open('$out/dp_chunks_$ch.json'))
print('chunks $ch one-rank', d['ms_per_step'], d['value'], d['check'])"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/single.json 2>$out/single.err
python -c "
import json
d=json.load(open('$out/single.json'))
print('single', d['ms_per_step'], d['value'], d['check'])"
