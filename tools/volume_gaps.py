#!/usr/bin/env python3
"""GPU busy / idle time of whole-volume runs from a rocprofv3 rocpd database of `bench.py --workload volume` (rocprofv3 --kernel-trace
-d DIR -o NAME -- python3 bench.py --workload volume --no-cpu-baseline): the dispatch timeline is cut into runs wherever the GPU sat
idle for more than `gap_ms` (model builds, synthetic-volume generation and file clean-up between the driver calls), and each run =
one volume through one driver reports wall, kernel-busy time, idle share and launches.  Order per dataset: warm-up, the pipelined
repeats (m3d.infer.infer_prm), then the serial driver (infer_prm_serial).
usage: tools/volume_gaps.py NAME_results.db [gap_ms]"""
import sqlite3
import sys


def main(db, gap_ms=12.0):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    nc = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (nc, kd, ks)).fetchall()
    runs, cur, prev_end = [], [], None
    for n, a, b in rows:
        if prev_end is not None and (a - prev_end) > gap_ms * 1e6 and cur:
            runs.append(cur)
            cur = []
        cur.append((n, a, b))
        prev_end = b if prev_end is None else max(prev_end, b)
    if cur:
        runs.append(cur)
    print("%d dispatches, %d runs (cut at idle gaps > %.0f ms); runs with fewer than 200 launches (weight packing, calibration) are not listed" % (len(rows), len(runs), gap_ms))
    print("%4s %10s %10s %8s %9s %9s  %s" % ("run", "wall ms", "busy ms", "idle %", "launches", "tiles", "largest gaps inside the run (us)"))
    k = 0
    for r in runs:
        if len(r) < 200:
            continue
        wall = (max(b for _, _, b in r) - r[0][1]) / 1e6
        # busy = union of the kernel intervals (a copy stream may overlap the compute stream)
        busy, end = 0.0, None
        gaps = []
        prev_name = ""
        for nm, a, b in r:
            if end is None or a >= end:
                if end is not None:
                    gaps.append(((a - end) / 1e3, prev_name, nm))
                busy += b - a
                end = b
            elif b > end:
                busy += b - end
                end = b
            prev_name = nm
        tiles = sum(1 for n, _, _ in r if "prm_select_peaks_kernel" in n)
        gaps.sort(reverse=True)
        print("%4d %10.2f %10.2f %8.1f %9d %9d  %s" % (k, wall, busy / 1e6, 100.0 * (1 - busy / 1e6 / wall), len(r), tiles,
                                                       " ".join("%.0f" % g[0] for g in gaps[:6])))
        if "-v" in sys.argv:
            short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]     # noqa: E731
            for g, a_, b_ in gaps[:8]:
                print("          gap %7.0f us between %-48s and %s" % (g, short(a_), short(b_)))
        k += 1


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-v" else 12.0)
