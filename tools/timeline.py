"""Prints the kernel timeline of the last training step found in a rocprofv3 kernel-trace CSV.
usage: python tools/timeline.py <kernel_trace.csv> [n_kernels_per_step_hint]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+), (\d), (\w+)>", n)
    if m:
        lay = ("N" if m.group(5) == "true" else "T") + ("T" if m.group(6) == "true" else "N")
        return "gemm %sx%s wk%s %s epi%s" % (m.group(1), m.group(2), m.group(4), lay, m.group(7))
    return n.split("(")[0][:40]


# one step = from one loss kernel to the next
idx = [i for i, r in enumerate(rows) if "xent_rows" in r["Kernel_Name"] or "softmax_xent" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
# start the step at the first kernel after the previous step's last update
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
queues = sorted(set(r["Queue_Id"] for r in step))
print("step of %d kernels, %.1f us, queues %s" % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e3, queues))
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = queues.index(r["Queue_Id"])
    print("%8.1f %8.1f %7.1f  %s%s  grid=%s" % (s / 1e3, e / 1e3, (e - s) / 1e3, "    " * q + "Q%d " % q, short(r["Kernel_Name"]),
                                              int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
