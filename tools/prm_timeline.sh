#!/bin/bash
WL=${1:-prm}
O=/root/repo/gpurun_out/prm_tl_$WL; mkdir -p $O; rm -rf /tmp/rp_prm
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_prm -o prm -- python3 /root/repo/bench.py --workload $WL --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/rp.err || { tail -5 $O/rp.err; exit 1; }
python3 - $(find /tmp/rp_prm -name "*_results.db" | head -1) > $O/timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
nc = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (nc, kd, ks)).fetchall()
# a step = one tile: the launches between two consecutive prm_seed launches (one per tile)
names = {}
for r in rows:
    names[r[0]] = names.get(r[0], 0) + 1
mark = [i for i, r in enumerate(rows) if "fc_x3_gemm_kernel" in r[0]]
# the box head runs once per tile (fc1 + fc2 = two launches): take every second marker
mark = mark[::2]
a, b = mark[-4], mark[-3]
seg = rows[a:b]
busy = sum(e - s for _, s, e in seg); span = rows[b][1] - rows[a][1]
print("one tile: span %.3f ms, kernels busy %.3f ms, idle %.3f ms, %d launches" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(seg)))
prev = None
for n, s, e in seg:
    gap = (s - prev) / 1e3 if prev else 0.0
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if (e - s) > 40e3 or gap > 8:
        print("%-80s %9.1f us  gap %7.1f%s" % (n[:80], (e - s) / 1e3, gap, "  <<<" if gap > 8 else ""))
    prev = max(prev or 0, e)
PY
cat $O/timeline.txt
