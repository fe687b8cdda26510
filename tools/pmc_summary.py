"""Prints per-kernel duration (kernel trace) and per-wave SQ counters of one tools/prof_gemm.sh run."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s n=%s avg=%.2f us min=%.2f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
for f in glob.glob(d + "/p1/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[r["Kernel_Name"]] = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    for k, c in agg.items():
        if "gemm_kernel" not in k:
            continue
        grid, wg, vg, ag, lds = meta[k]
        waves = grid // 64
        m = {n: sum(v) / len(v) for n, v in c.items()}
        print(k[:100])
        print("  waves=%d vgpr=%s agpr=%s lds=%s" % (waves, vg, ag, lds))
        for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            print("  %-28s %10.0f cyc/wave" % (n, 4 * m[n] / waves))
        print("  %-28s %10.0f cyc/wave" % ("MFMA busy", m["SQ_VALU_MFMA_BUSY_CYCLES"] / waves))
        print("  %-28s %10.0f (per SE)" % ("SQ_BUSY_CYCLES", m["SQ_BUSY_CYCLES"] / 32))
