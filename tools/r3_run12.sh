#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run12; mkdir -p $O
for cfg in "40 5 10" "40 5 40" "10 3 10" "100 2 10" "5 1 10"; do
set -- $cfg
M3D_BENCH_PROBE_STEPS=$3 timeout -k 10 400 python bench.py --steps $1 --warmup $2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O/bench.json "$cfg" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps warmup probed = %s: serial %.3f ms | sustained %.3f | fc1 %.3f conv2b %.3f" % (sys.argv[2], r["ms_per_step"], r.get("sustained", {}).get("ms_per_step", 0),
      r["rooflines"]["fc1"].get("kernel_ms", 0), r["rooflines"]["conv2b"].get("kernel_ms", 0)))
PY
done
