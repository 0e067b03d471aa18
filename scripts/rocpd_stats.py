#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table (what --stats prints as CSV).
usage: rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
    rows = cur.execute("select * from kernels").fetchall()
    name_i = cols.index("name") if "name" in cols else cols.index("kernel_name")
    s_i, e_i = cols.index("start"), cols.index("end")
    stat = {}
    for r in rows:
        d = stat.setdefault(r[name_i], [])
        d.append((r[e_i] - r[s_i]) / 1e3)   # us
    total = sum(sum(v) for v in stat.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(stat.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (k[:110], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / total))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
