#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME -- prog` writes NAME_results.db):
prints / writes the same columns as `--stats`' kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs)."""
import csv
import sqlite3
import sys


def main(db, out=None, top=40):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)).fetchall()
    agg = {}
    for n, a, b in rows:
        agg.setdefault(n, []).append(b - a)
    tot = sum(sum(v) for v in agg.values())
    tab = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v)) for n, v in agg.items()), key=lambda r: -r[2])
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in tab:
                w.writerow(r)
    for r in tab[:top]:
        print("%-100s calls %5d  avg %10.1f us  %5.1f%%  min %9.1f  max %9.1f" % (r[0][:100], r[1], r[3] / 1e3, r[4], r[5] / 1e3, r[6] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
