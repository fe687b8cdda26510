"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share,
with template arguments shortened.  usage: python tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s*\[clone .*\]", "", name)
    return name[:150]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)), max(lds_size), "
                      "max(vgpr_count), max(accum_vgpr_count) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | blocks | LDS B | VGPR | AGPR |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for r in rows:
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f | %d | %d | %d | %d |" %
                     (short(r[0]), r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7],
                      r[8], r[9]))
    text = "\n".join(lines) + "\n\ntotal kernel time %.3f ms over %d dispatches\n" % (total / 1e6, sum(r[1] for r in rows))
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(text)
    print(text)


if __name__ == "__main__":
    main()
