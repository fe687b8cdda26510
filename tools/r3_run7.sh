set -e
out=gpurun_out/r3_t
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -k "chain or gemm or affine" > $out/pytest7.txt 2>&1 || { tail -60 $out/pytest7.txt; exit 1; }
tail -3 $out/pytest7.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
