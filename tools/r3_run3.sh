set -e
out=gpurun_out/r3_bench
mkdir -p $out
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/default.json 2> $out/default.err || { tail -30 $out/default.err; exit 1; }
echo "default run took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3_bench/default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"])
for e in d["secondary"]:
    print(e["workload"], e["value"], e["ms_per_step"], e["frac"], e["frac_gemm_launches"], e["check"], (e.get("cpu_baseline") or {}).get("value"), (e.get("cpu_baseline") or {}).get("sample"))
PY
