set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest_all.txt 2>&1 || { tail -60 $out/pytest_all.txt; exit 1; }
tail -3 $out/pytest_all.txt
for wl in rm swbd; do
  echo -n "$wl: "; timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 400 --warmup 40 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['pct_fp32_mfma_peak_end_to_end'], d['check'])"
done
