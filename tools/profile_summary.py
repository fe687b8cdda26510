"""Markdown summary of one tools/profile_bench.sh run: per-kernel stats from the kernel trace,
HBM bytes per launch (FETCH_SIZE doubled as gfx950 needs, + WRITE_SIZE, in KiB units) and the
MFMA-busy share of the GEMM kernels.  usage: python tools/profile_summary.py <dir> [out.md]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n[:70]


def one(pattern):
    f = glob.glob(d + pattern, recursive=True)
    return f[0] if f else None


lines = []
f = one("/kt/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines += ["### kernel trace (`rocprofv3 --kernel-trace --stats`, begin-to-end stamps of every launch of the process, set-up and timed-dispatch passes included)", "",
              "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append("| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.1f |" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    g = [r for r in rows if "gemm_" in r["Name"]]
    gc, gt = sum(int(r["Calls"]) for r in g), sum(float(r["TotalDurationNs"]) for r in g)
    lines += ["", "all GEMM kernels (`gemm_kernel` instantiations, `gemm_update_group_kernel`): %d launches, %.3f ms, average %.2f us per launch" % (gc, gt / 1e6, gt / gc / 1e3 if gc else 0), ""]


def pmc(sub):
    f = one("/%s/**/*counter_collection.csv" % sub)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg, None


fa, _ = pmc("fetch")
wa, _ = pmc("write")
if fa or wa:
    lines += ["### HBM-side bytes per launch (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes; KiB counters; "
              "FETCH_SIZE doubled: gfx950 tallies 128-B read requests at 64 B)", "",
              "| kernel | launches | read MB | written MB |", "|---|---:|---:|---:|"]
    tr = tw = n = 0
    for k in sorted(set(fa) | set(wa)):
        fr = fa.get(k, {}).get("FETCH_SIZE", [])
        wr = wa.get(k, {}).get("WRITE_SIZE", [])
        rd = 2 * 1024 * sum(fr) / max(len(fr), 1) / 1e6
        wt = 1024 * sum(wr) / max(len(wr), 1) / 1e6
        lines.append("| `%s` | %d | %.2f | %.2f |" % (k, max(len(fr), len(wr)), rd, wt))
        if "gemm_" in k:
            tr += 2 * 1024 * sum(fr)
            tw += 1024 * sum(wr) * (len(fr) / max(len(wr), 1))
            n += len(fr)
    if n:
        lines += ["", "all GEMM launches: %.2f MB read + %.2f MB written = %.2f MB per launch on average" %
                  (tr / n / 1e6, tw / n / 1e6, (tr + tw) / n / 1e6), "", "GEMM_TRAFFIC_BYTES_PER_LAUNCH=%d" % ((tr + tw) / n), ""]
# SQ pass: one row per (kernel, grid size) -- the same instantiation serves layers of different sizes
f = one("/sq/**/*counter_collection.csv")
if f:
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "gemm" in k:
            sq[(k, int(r["Grid_Size"]) // 64)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["### SQ counters of the GEMM kernels, per launch shape (one `--pmc` pass; kernels serialised by the profiler)", "",
              "MFMA pipe busy = `SQ_VALU_MFMA_BUSY_CYCLES` (busy cycles summed over all SIMDs; 64 per `v_mfma_f32_32x32x2_f32`) / "
              "(1024 SIMDs x the launch's length, taken as `SQ_BUSY_CYCLES` / 32 shader engines): the share of the chip's matrix-core "
              "cycles this launch used.  The per-wave columns are averages over the launch's waves, in cycles (`SQ_*` x 4); a wave "
              "that shares its SIMD with n - 1 others can be MFMA-busy 1/n of its own lifetime at most.", "",
              "| kernel | waves | launches | launch cycles | MFMA pipe busy % | wave lifetime | wave's own MFMA | parked (s_waitcnt/barrier) | issue-stall | LDS issue-stall |",
              "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for (k, w) in sorted(sq):
        m = {c: sum(v) / len(v) for c, v in sq[(k, w)].items()}
        n = len(sq[(k, w)]["SQ_WAVE_CYCLES"])
        launch = m["SQ_BUSY_CYCLES"] / 32.0
        lines.append("| `%s` | %d | %d | %.0f | %.1f | %.0f | %.0f | %.0f | %.0f | %.0f |" % (
            k, w, n, launch, 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * launch), 4 * m["SQ_WAVE_CYCLES"] / w,
            m["SQ_VALU_MFMA_BUSY_CYCLES"] / w, 4 * m["SQ_WAIT_ANY"] / w, 4 * m["SQ_WAIT_INST_ANY"] / w, 4 * m["SQ_WAIT_INST_LDS"] / w))
    lines.append("")
bench = [l for l in open(d + "/kt.log") if l.startswith("{")]
if bench:
    lines += ["### bench.py line of the traced run (slower than an untraced run: the tracer lengthens dispatch gaps)", "", "```", bench[-1].strip(), "```", ""]
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
