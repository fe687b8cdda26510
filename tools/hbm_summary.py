#!/usr/bin/env python3
"""Merges tools/hbm_kernels.py's timing table with its rocprofv3 FETCH_SIZE / WRITE_SIZE passes into a markdown table.
usage: python tools/hbm_summary.py <timing.jsonl> <fetch_dir> <write_dir> <fetch_cases.jsonl> <write_cases.jsonl>"""
import csv
import glob
import json
import sys


def cases(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and not r["Kernel_Name"].startswith("__amd_rocclr")]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


timing = cases(sys.argv[1])
out = []
for counter, d, cfile in (("FETCH_SIZE", sys.argv[2], sys.argv[4]), ("WRITE_SIZE", sys.argv[3], sys.argv[5])):
    disp = per_dispatch(d, counter)
    pos = 0
    vals = []
    for c in cases(cfile):
        chunk = disp[pos:pos + c["launches"]]
        pos += c["launches"]
        assert all(c["kernel"] in k for k, _ in chunk), (c, chunk)
        vals.append(sum(v for _, v in chunk) / c.get("calls", len(chunk)) * 1024.0)   # KiB counters, per call
    out.append(vals)
print("| kernel (entry point) | shape | us / launch | algorithmic MB | GB/s | % of 8 TB/s | read MB (FETCH_SIZE x 2) | written MB (WRITE_SIZE) |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
for t, fe, wr in zip(timing, out[0], out[1]):
    print("| %s | %d x %d | %.2f | %.1f | %.0f | %.1f | %.1f | %.1f |" % (t["case"], t["shape"][0], t["shape"][1], t["us"], t["bytes"] / 1e6,
                                                                  t["GBps"], 100 * t["frac_of_8TBps"], 2 * fe / 1e6, wr / 1e6))
