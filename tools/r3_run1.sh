set -e
out=gpurun_out/r3_dp
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_dist.py -x -q --timeout 120 --timeout-method thread > $out/pytest_dist.txt 2>&1 || { tail -40 $out/pytest_dist.txt; exit 1; }
tail -3 $out/pytest_dist.txt
for ch in 3 2 1; do
  KALDI_AMD_DP_CHUNKS=$ch KALDI_AMD_DP_EXCHANGE=shard KALDI_BENCH_LOOPBACK_DP=8 KALDI_BENCH_LOOPBACK_ONE_RANK=1 timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/dp_chunks_$ch.json 2> $out/dp_chunks_$ch.err
  python -c "
import json
d=json.load(open('$out/dp_chunks_$ch.json'))
print('chunks $ch one-rank', d['ms_per_step'], d['value'], d['check'])"
done
KALDI_AMD_DP_EXCHANGE=shard KALDI_BENCH_LOOPBACK_DP=8 timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/dp_exact.json 2> $out/dp_exact.err
python -c "
import json
d=json.load(open('$out/dp_exact.json'))
print('exact 8', d['ms_per_step'], d['value'], d['check'])"
CHUNKS=3 bash tools/r3_dp_trace.sh
