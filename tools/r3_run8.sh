set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_nnet.py tests/test_gpu_dist.py -x -q --timeout 120 --timeout-method thread -k "sample or rbm or sharded_exchange or rand or hybrid" > $out/pytest8.txt 2>&1 || { tail -60 $out/pytest8.txt; exit 1; }
tail -3 $out/pytest8.txt
for wl in rbm-gauss-bern rbm-bern-bern; do
  for env in "" KALDI_AMD_UNFUSED_SAMPLING=1; do
    echo -n "$wl $env: "; env $env timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps 400 --warmup 40 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['pct_fp32_mfma_peak_end_to_end'])"
  done
done
