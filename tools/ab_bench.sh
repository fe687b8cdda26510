#!/bin/bash
# A/B on one GPU box: bench.py with the built device library (base) and with tools/dbg/libkaldi_hip_alt.so (alt),
# alternating; the alt library is first checked with the GEMM parity tests.  usage: tools/ab_bench.sh [rounds]
set -e
rounds=${1:-2}
cp kaldi_amd/lib/libkaldi_hip.so /tmp/base.so
cp tools/dbg/libkaldi_hip_alt.so kaldi_amd/lib/libkaldi_hip.so
timeout -k 10 150 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or affine or rbm" 2>&1 | tail -1
for i in $(seq $rounds); do
  cp /tmp/base.so kaldi_amd/lib/libkaldi_hip.so
  echo -n "base: "; timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c140-165
  cp tools/dbg/libkaldi_hip_alt.so kaldi_amd/lib/libkaldi_hip.so
  echo -n "alt:  "; timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c140-165
done
cp /tmp/base.so kaldi_amd/lib/libkaldi_hip.so
