#!/bin/bash
# The data-parallel training step rehearsed on ONE GPU with the loopback communicator (no link traffic): one bench.py
# line per exchange, number of stand-in ranks and stand-in mode; then a kernel trace of the default (shard, 8 ranks,
# one rank's work).  Writes gpurun_out/dp_rehearsal/{lines.jsonl,table.md,timeline.txt}.
set -e
ROOT=$(pwd)
out=$ROOT/gpurun_out/dp_rehearsal
mkdir -p $out
: > $out/lines.jsonl
run() {  # exchange ranks one_rank chunks
  local tag="$1 $2 $3 $4"
  env KALDI_AMD_DP_EXCHANGE=$1 KALDI_BENCH_LOOPBACK_DP=$2 ${3:+KALDI_BENCH_LOOPBACK_ONE_RANK=1} ${4:+KALDI_AMD_DP_CHUNKS=$4} \
    timeout -k 10 180 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/line.json 2> $out/line.err
  python3 - "$1" "$2" "${3:-}" "${4:-}" $out/line.json >> $out/lines.jsonl <<'PY'
import json, sys
d = json.load(open(sys.argv[5]))
print(json.dumps({"exchange": sys.argv[1], "stand_in_ranks": int(sys.argv[2]), "stand_in": "one rank's work" if sys.argv[3] else "all ranks' work",
                  "chunks": sys.argv[4] or "default", "ms_per_step": d["ms_per_step"], "frames_per_s": d["value"], "check": d["check"]}))
PY
  echo "$tag: $(tail -1 $out/lines.jsonl)"
}
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/single.json 2> $out/single.err
for n in 2 4 8; do
  for ex in shard sum gather; do run $ex $n "" ""; done
  run shard $n 1 ""
done
for ch in 1 2 8; do run shard 8 1 $ch; done
cd /tmp && export TMPDIR=/tmp
KALDI_AMD_DP_EXCHANGE=shard KALDI_BENCH_LOOPBACK_DP=8 KALDI_BENCH_LOOPBACK_ONE_RANK=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/kt.log 2>&1
python3 $ROOT/tools/timeline.py $out/kt/kt_kernel_trace.csv > $out/timeline.txt
cd $ROOT
python3 - $out <<'PY'
import json, sys
out = sys.argv[1]
single = json.load(open(out + "/single.json"))
rows = [json.loads(l) for l in open(out + "/lines.jsonl")]
with open(out + "/table.md", "w") as f:
    f.write("| exchange | stand-in ranks | the stand-in does | chunks | ms per step | frames/s per GPU |\n|---|---:|---|---|---:|---:|\n")
    f.write("| (single GPU, no exchange) | 1 | - | - | %.4f | %.0f |\n" % (single["ms_per_step"], single["value"]))
    for r in rows:
        f.write("| %s | %d | %s | %s | %.4f | %.0f |\n" % (r["exchange"], r["stand_in_ranks"], r["stand_in"], r["chunks"], r["ms_per_step"], r["frames_per_s"]))
print(open(out + "/table.md").read())
PY
