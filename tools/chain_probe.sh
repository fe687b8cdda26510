#!/bin/bash
# A/B on one GPU box: the training step with the built device library against the chain-probe builds
# (kaldi_amd/lib/libkaldi_hip_probe{1,2}.so: make alt ALT=-DKH_CHAIN_PROBE=n, copied there), alternating.
set -e
out=gpurun_out/chain_probe
mkdir -p $out
cp kaldi_amd/lib/libkaldi_hip.so /tmp/base.so
line() { timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --workload $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
pk={k['kernel'][:28]:k['avg_launch_us'] for k in d['roofline']['per_kernel']}
print(d['ms_per_step'], pk)"; }
for i in 1 2; do
  for lib in base probe1 probe2; do
    if [ $lib = base ]; then cp /tmp/base.so kaldi_amd/lib/libkaldi_hip.so; else cp kaldi_amd/lib/libkaldi_hip_$lib.so kaldi_amd/lib/libkaldi_hip.so; fi
    for wl in swbd rm; do echo -n "$lib $wl: "; line $wl; done
  done
done | tee $out/ab.txt
cp /tmp/base.so kaldi_amd/lib/libkaldi_hip.so
