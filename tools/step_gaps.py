#!/usr/bin/env python3
"""Timeline of one steady-state detection step from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...):
every kernel of the step with its duration and the idle gap before it; totals of busy / idle time per step.
A step starts at each norm1_sum_kernel launch; the steps printed are the last `n` of the run (the sustained loop, no event probes).
usage: tools/step_gaps.py NAME_results.db [n_steps_to_average]"""
import sqlite3
import sys


def main(db, nsteps=40):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)).fetchall()
    starts = [i for i, r in enumerate(rows) if "norm1_sum_kernel" in r[0]]
    if len(starts) < nsteps + 2:
        nsteps = max(1, len(starts) - 2)
    sel = starts[-(nsteps + 1):]
    steps = [rows[sel[i]:sel[i + 1]] for i in range(nsteps)]
    lens = [s[-1][2] - s[0][1] for s in steps]
    period = [(rows[sel[i + 1]][1] - rows[sel[i]][1]) for i in range(nsteps)]
    busy = [sum(b - a for _, a, b in s) for s in steps]
    print("%d steps: period %.3f ms (median), kernels busy %.3f ms, idle inside the step %.3f ms, launches per step %d" %
          (nsteps, sorted(period)[nsteps // 2] / 1e6, sorted(busy)[nsteps // 2] / 1e6,
           (sorted(period)[nsteps // 2] - sorted(busy)[nsteps // 2]) / 1e6, len(steps[-1])))
    # one representative step: the one with the median period
    k = sorted(range(nsteps), key=lambda i: period[i])[nsteps // 2]
    s = steps[k]
    prev_end = None
    print("%-72s %9s %9s" % ("kernel", "dur us", "gap us"))
    for n, a, b in s:
        gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-72s %9.1f %9.1f%s" % (n[:72], (b - a) / 1e3, gap, "   <<<" if gap > 8 else ""))
        prev_end = max(prev_end or 0, b)
    nxt = rows[sel[k + 1]][1]
    print("%-72s %9s %9.1f" % ("(next step's first kernel)", "", (nxt - prev_end) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
